"""Image directory -> boxes / scores / classes: the MI355X counterpart of the reference's
inference.py, same public names and argument order:

    read_image(img_path)                    inference.py:28
    get_classes(labels_path)                inference.py:56
    read_anchors(file_path)                 inference.py:70
    predict(output_nodes, anchors, num_classes, input_shape, image_shape)   inference.py:88
    run_inference(img_path, output_dir, args)                               inference.py:170
    CLI:  python -m yolo_amd.inference <img_dir> <out_dir> [--darknet_model 1]   inference.py:20-24

predict() takes evaluated head tensors (what `sess.run(output_nodes)` would give) and runs
the fused decode + per-class NMS kernels.  Drawing (inference.py:284-325) follows the reference (class
colours, outline thickness, label placement) with whatever TrueType font PIL finds, since the reference's
font file is not shipped; PIL is used for image decode in place of cv2 (absent here).
"""
import argparse
import os
from time import time

import numpy as np
import torch

from . import config
from .build_model import load_weights, yolo
from .postprocess import PostProcessor
from .utils.utils import letterbox_gpu, resize_image  # noqa: F401  (resize_image: reference API, host path)

parser = argparse.ArgumentParser(description="Run inference using darknet converted model")
parser.add_argument("img_path", help="Path for running inference on a single image or multiple images")
parser.add_argument("output_path", help="Output Path to save the results")
parser.add_argument("--darknet_model", help="Weather to use pre-trained darknet yolov3")
parser.add_argument("--weights", default=None, help="darknet .weights file (default: config.yolov3_weights_path)")
parser.add_argument("--cfg", default=None, help="darknet .cfg file (default: config.yolov3_cfg_path)")
parser.add_argument("--input_shape", type=int, default=None, help="network input size (default: config.input_shape)")
parser.add_argument("--dtype", default=None, choices=["bf16", "fp32"])
parser.add_argument("--no_splitk", action="store_true", help="do not split small launches over k (bit-identical to the batched path)")
parser.add_argument("--batch", type=int, default=1, help="images per step (default 1, the reference's mode): the directory is processed in "
                    "chunks of this many images; every image gets the result it gets alone (with --no_splitk bit for bit)")
parser.add_argument("--graph", action="store_true", help="replay one captured HIP graph per image instead of launching every kernel from the host")


def read_image(img_path):
    """RGB uint8 arrays of every image in a directory, sorted by name (inference.py:28-52)."""
    from PIL import Image
    assert img_path is not None, "Image path required for making inference"
    if not os.path.exists(img_path):
        print("Path does not exists!!")
        return None
    names = sorted(os.listdir(img_path)) if os.path.isdir(img_path) else [None]
    print("Reading {} images".format(len(names)))
    images = []
    for name in names:
        path = img_path if name is None else os.path.join(img_path, name)
        images.append(np.asarray(Image.open(path).convert("RGB")))
    print("Read {} images".format(len(names)))
    return images


def get_classes(labels_path):
    with open(labels_path) as f:
        return [c.strip() for c in f.readlines()]


def read_anchors(file_path):
    """`w h` per line -> float array [n,2] (inference.py:70-84)."""
    anchors = []
    with open(file_path, "r") as fh:
        for line in fh.read().splitlines():
            if not line.strip():
                continue
            w, h = line.split()
            anchors.append([float(w), float(h)])
    return np.asarray(anchors)


_POST = {}


def _postprocessor(anchors, num_classes, input_shape, with_rank=False):
    key = (np.asarray(anchors, dtype=np.float32).tobytes(), num_classes, tuple(input_shape), config.score_threshold,
           config.nms_threshold, config.max_boxes, with_rank)
    pp = _POST.get(key)
    if pp is None:
        pp = PostProcessor(anchors, num_classes, input_shape, config.score_threshold, config.nms_threshold,
                           config.max_boxes, with_rank=with_rank)
        _POST[key] = pp
    return pp


def predict_batch(output_nodes, anchors, num_classes, input_shape, image_shape):
    """Batch form: B independent batch-1 reference runs (SURVEY 8a batch note).
    image_shape: (h, w) or [B,2].  Returns postprocess.Detections (padded, on the GPU)."""
    return _postprocessor(anchors, num_classes, input_shape)(list(output_nodes), image_shape)


def detect_batch(output_nodes, images, anchors, num_classes, input_shape, image_shape, chunk=None, want_heads=False):
    """Images -> detections in one go: what the reference's `sess.run(predict(output_nodes, ...), {input: images})`
    evaluates (inference.py:88-167, 277).  output_nodes: what `yolo()` returned (or a build_model.Network).  The decode
    of each scale runs inside its head convolution where the plan allows it, so the fp32 heads are not read back (nor
    written, unless want_heads); otherwise this is `predict_batch(output_nodes(images), ...)`.  Same results either way.
    Returns postprocess.Detections, or (Detections, heads) with want_heads."""
    net = getattr(output_nodes, "net", output_nodes)
    det, heads = _postprocessor(anchors, num_classes, input_shape).detect(net, images, image_shape, chunk=chunk,
                                                                          want_heads=want_heads)
    return (det, heads) if want_heads else det


def predict(output_nodes, anchors, num_classes, input_shape, image_shape):
    """Reference contract (inference.py:88-167): boxes [M,4] (y_min,x_min,y_max,x_max) fp32,
    scores [M] fp32, classes [M] int32; class-major, within a class in NMS selection order.
    Thresholds come from config (score_threshold, nms_threshold, max_boxes; inference.py:102-104).
    Batch 1 returns the triple; batch B returns a list of B triples."""
    det = predict_batch(output_nodes, anchors, num_classes, input_shape, image_shape)
    counts = det.count.cpu().numpy()
    out = [(det.boxes[b, :m].clone(), det.scores[b, :m].clone(), det.classes[b, :m].clone())
           for b, m in enumerate(counts)]
    return out[0] if len(out) == 1 else out


def label_lines(out_boxes, out_scores, out_classes, image_hw):
    """The labels/<stem>.txt lines (inference.py:291-308): rows in reverse order,
    `left,top,right,bottom,class,score`, floor(v+0.5), clipped to the image."""
    h, w = image_hw
    lines = []
    for i in reversed(range(len(out_classes))):
        top, left, bottom, right = out_boxes[i]
        top = max(0, np.floor(top + 0.5).astype(np.int32))
        left = max(0, np.floor(left + 0.5).astype(np.int32))
        bottom = min(h, np.floor(bottom + 0.5).astype(np.int32))
        right = min(w, np.floor(right + 0.5).astype(np.int32))
        lines.append(str(left) + "," + str(top) + "," + str(right) + "," + str(bottom) + "," + str(out_classes[i])
                     + "," + str(out_scores[i]))
    return lines


def class_colors(num):
    """One RGB colour per class, as inference.py:209-215: evenly spaced hues, shuffled with the fixed seed
    10101 so that adjacent classes differ and runs agree."""
    import colorsys
    import random
    colors = [colorsys.hsv_to_rgb(x / num, 1., 1.) for x in range(num)]
    colors = [(int(r * 255), int(g * 255), int(b * 255)) for r, g, b in colors]
    rnd = random.Random(10101)              # the reference seeds the global generator and resets it afterwards
    rnd.shuffle(colors)
    return colors


def _label_font(image_width):
    """The reference loads ./font/FiraMono-Medium.otf at size floor(0.01*w + 0.5) (inference.py:285-286); that
    file is not shipped here, so any TrueType font PIL can find is used at that size, else PIL's built-in."""
    from PIL import ImageFont
    size = max(int(np.floor(1e-2 * image_width + 0.5)), 8)
    for name in (getattr(config, "font_path", None), "DejaVuSansMono.ttf", "DejaVuSans.ttf"):
        if not name:
            continue
        try:
            return ImageFont.truetype(name, size=size)
        except (OSError, ValueError):
            continue
    return ImageFont.load_default()


def draw_detections(image, out_boxes, out_scores, out_classes, class_names, colors=None):
    """Annotated copy of `image` (uint8 RGB array) as inference.py:284-321 draws it: per detection, in reverse
    order, a box outline (h+w)//500 pixels thick in the class colour and a filled label `name score` above
    the box (inside it when there is no room).  Returns a PIL image."""
    from PIL import Image, ImageDraw
    colors = colors or class_colors(len(class_names))
    h, w = image.shape[:2]
    font = _label_font(w)
    thickness = (h + w) // 500
    img = Image.fromarray(np.asarray(image).astype("uint8"), mode="RGB")
    draw = ImageDraw.Draw(img)
    for i, c in reversed(list(enumerate(out_classes))):
        label = "{} {:.4f}".format(class_names[c], out_scores[i])
        x0, y0, x1, y1 = draw.textbbox((0, 0), label, font=font)
        label_size = np.array([x1 - x0, y1 - y0])
        top, left, bottom, right = out_boxes[i]
        top = max(0, int(np.floor(top + 0.5)))
        left = max(0, int(np.floor(left + 0.5)))
        bottom = min(h, int(np.floor(bottom + 0.5)))
        right = min(w, int(np.floor(right + 0.5)))
        origin = np.array([left, top - label_size[1]]) if top - label_size[1] >= 0 else np.array([left, top + 1])
        for j in range(thickness):
            if right - j >= left + j and bottom - j >= top + j:
                draw.rectangle([left + j, top + j, right - j, bottom - j], outline=colors[c])
        draw.rectangle([tuple(origin), tuple(origin + label_size)], fill=colors[c])
        draw.text(tuple(origin), label, fill=(0, 0, 0), font=font)
    return img


def run_inference(img_path, output_dir, args):
    """inference.py:170-331: every image of img_path -> output_dir/labels/<stem>.txt and the annotated image
    under output_dir/images, FPS excluding the first step.  The reference runs one image per step
    (inference.py:243); `args.batch` = N runs the directory in chunks of N images per step (the last chunk may be
    shorter).  A batch is B independent batch-1 runs and the batched plan gives an image the same bits
    whatever the batch size, so the files do not depend on N (tests: byte-identical to batch 1 run with --no_splitk;
    batch 1 splits small launches over k by default, which adds the partial sums in another order)."""
    for d in (output_dir, os.path.join(output_dir, "images"), os.path.join(output_dir, "labels")):
        if not os.path.exists(d):
            os.mkdir(d)
    file_names = sorted(os.listdir(img_path))
    images_batch = read_image(img_path)
    class_names = get_classes(config.classes_path)
    cfg_path = getattr(args, "cfg", None) or config.yolov3_cfg_path
    tiny = "tiny" in os.path.basename(cfg_path)
    anchors = read_anchors(config.tiny_anchors_path if tiny else config.anchors_path)
    num_classes = config.num_classes
    size = getattr(args, "input_shape", None) or config.input_shape
    input_shape = (size, size)
    batch = max(1, int(getattr(args, "batch", 1) or 1))

    # one image per step: launches with few output tiles are split over k (1.6 -> ~1 ms per 608 image)
    output_nodes = yolo(input_images=None, is_training=False, config_path=cfg_path, num_classes=num_classes,
                        dtype=getattr(args, "dtype", None), splitk=batch == 1 and not getattr(args, "no_splitk", False))
    weights = getattr(args, "weights", None) or config.yolov3_weights_path
    if getattr(args, "darknet_model", None) is None and getattr(args, "weights", None) is None:
        raise NotImplementedError("only Darknet .weights are supported (pass --darknet_model 1); the TF-checkpoint "
                                  "branch of the reference (inference.py:258-266) is out of scope")
    load_weights(output_nodes, weights)      # FileNotFoundError if absent: nothing is downloaded

    colors = class_colors(len(class_names))
    total_time_pred = []
    results = []
    # --graph captures the ~80 launches of a step once as a HIP graph and replays it per step
    # (graph.GraphedDetector): one submission per step, same results; measured latency at batch 1 is the same
    # 1.58 ms at 608 (the step is bound by its chain of small kernels, not the host).
    graphed = None
    if getattr(args, "graph", False):
        from .graph import GraphedDetector
        graphed = GraphedDetector(output_nodes.net, anchors, num_classes, input_shape, batch=batch,
                                  score_threshold=config.score_threshold, iou_threshold=config.nms_threshold,
                                  max_boxes=config.max_boxes)
    staged = None if graphed is not None else torch.zeros((batch,) + input_shape + (3,), dtype=torch.float32,
                                                          device=output_nodes.net.device)
    for first in range(0, len(images_batch), batch):
        chunk = images_batch[first:first + batch]
        n = len(chunk)
        hws = [[im.shape[0], im.shape[1]] for im in chunk]
        for im in chunk:
            print("Image height: {}\tImage width: {}".format(im.shape[0], im.shape[1]))
        # resize_image + /255. (inference.py:238-243) on the device: same pixels as the host path
        target = graphed.images if graphed is not None else staged
        for j, im in enumerate(chunk):
            letterbox_gpu(im, input_shape, out=target[j])
        if graphed is not None:
            # a short last chunk replays the full-batch graph: the unused slots keep the previous chunk's images,
            # whose results are ignored (images are independent)
            tick = time()
            graphed(hws + [hws[-1]] * (batch - n))
            triples = [graphed.image(j) for j in range(n)]
        else:
            tick = time()
            det = detect_batch(output_nodes, staged[:n], anchors, num_classes, input_shape, np.asarray(hws))
            counts = det.count.cpu().numpy()
            triples = [(det.boxes[j, :m].cpu().numpy(), det.scores[j, :m].cpu().numpy(), det.classes[j, :m].cpu().numpy())
                       for j, m in enumerate(counts)]
        tock = time()
        total_time_pred.append((tock - tick, n))
        for j, (image, (out_boxes, out_scores, out_classes)) in enumerate(zip(chunk, triples)):
            x = first + j
            print("Found {} boxes for {} in {}sec".format(len(out_boxes), "img", (tock - tick) / n))
            lines = label_lines(out_boxes, out_scores, out_classes, image.shape[:2])
            for i, c in reversed(list(enumerate(out_classes))):
                print("{} {:.4f}".format(class_names[c], out_scores[i]))
            with open(os.path.join(output_dir, "labels", file_names[x].split(".")[0] + ".txt"), "w") as fh:
                fh.write("".join(l + "\n" for l in lines))
            draw_detections(image, out_boxes, out_scores, out_classes, class_names, colors).save(
                os.path.join(output_dir, "images", file_names[x]), compress_level=1)
            results.append((out_boxes, out_scores, out_classes))
    if len(total_time_pred) > 1:
        t = sum(dt for dt, _ in total_time_pred[1:])
        n = sum(k for _, k in total_time_pred[1:])
        print("FPS of model with post processing over {} images is {}".format(n, n / t))
    return results


def main(args):
    os.environ.setdefault("HIP_VISIBLE_DEVICES", str(config.gpu_num))
    run_inference(args.img_path, args.output_path, args)


if __name__ == "__main__":
    main(parser.parse_args())
