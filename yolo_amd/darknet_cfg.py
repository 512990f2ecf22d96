"""Generators for the Darknet cfg topologies the path is benchmarked on.

The cfg is data the reference reads (darknet_data/yolov3.cfg, yolov3-tiny.cfg).  The
files packaged under yolo_amd/darknet_data are written by this module and carry only
the keys the inference path consumes (build_model.py:116-121,185,204-205,219-220,232).
`tests/test_cfg.py` checks them section by section against the reference's files when
/root/reference is present.
"""
import os

COCO_ANCHORS = [(10, 13), (16, 30), (33, 23), (30, 61), (62, 45), (59, 119), (116, 90), (156, 198), (373, 326)]
TINY_ANCHORS = [(10, 14), (23, 27), (37, 58), (81, 82), (135, 169), (344, 319)]


def _conv(filters, size, stride=1, bn=True, act="leaky"):
    s = "[convolutional]\n"
    if bn:
        s += "batch_normalize=1\n"
    return s + "filters=%d\nsize=%d\nstride=%d\npad=1\nactivation=%s\n\n" % (filters, size, stride, act)


def _yolo(mask, anchors, classes):
    a = ",  ".join("%d,%d" % wh for wh in anchors)
    return "[yolo]\nmask = %s\nanchors = %s\nclasses=%d\nnum=%d\n\n" % (",".join(map(str, mask)), a, classes, len(anchors))


def _net(size):
    return "[net]\nbatch=1\nsubdivisions=1\nwidth=%d\nheight=%d\nchannels=3\n\n" % (size, size)


def yolov3_cfg(classes=80, size=608):
    out = [_net(size), _conv(32, 3)]
    for filters, blocks in ((64, 1), (128, 2), (256, 8), (512, 8), (1024, 4)):
        out.append(_conv(filters, 3, stride=2))
        for _ in range(blocks):
            out.append(_conv(filters // 2, 1))
            out.append(_conv(filters, 3))
            out.append("[shortcut]\nfrom=-3\nactivation=linear\n\n")
    nout = 3 * (classes + 5)
    for k, (filters, mask, route_to) in enumerate(((512, (6, 7, 8), 61), (256, (3, 4, 5), 36), (128, (0, 1, 2), None))):
        for _ in range(3):
            out.append(_conv(filters, 1))
            out.append(_conv(filters * 2, 3))
        out.append(_conv(nout, 1, bn=False, act="linear"))
        out.append(_yolo(mask, COCO_ANCHORS, classes))
        if route_to is not None:
            out.append("[route]\nlayers = -4\n\n")
            out.append(_conv(filters // 2, 1))
            out.append("[upsample]\nstride=2\n\n")
            out.append("[route]\nlayers = -1, %d\n\n" % route_to)
    return "".join(out)


def yolov3_tiny_cfg(classes=80, size=416):
    out = [_net(size)]
    for filters in (16, 32, 64, 128, 256):
        out.append(_conv(filters, 3))
        out.append("[maxpool]\nsize=2\nstride=2\n\n")
    out.append(_conv(512, 3))
    out.append("[maxpool]\nsize=2\nstride=1\n\n")
    out.append(_conv(1024, 3))
    out.append(_conv(256, 1))
    out.append(_conv(512, 3))
    nout = 3 * (classes + 5)
    out.append(_conv(nout, 1, bn=False, act="linear"))
    out.append(_yolo((3, 4, 5), TINY_ANCHORS, classes))
    out.append("[route]\nlayers = -4\n\n")
    out.append(_conv(128, 1))
    out.append("[upsample]\nstride=2\n\n")
    out.append("[route]\nlayers = -1, 8\n\n")
    out.append(_conv(256, 3))
    out.append(_conv(nout, 1, bn=False, act="linear"))
    out.append(_yolo((0, 1, 2), TINY_ANCHORS, classes))
    return "".join(out)


COCO_CLASSES = """person bicycle car motorbike aeroplane bus train truck boat traffic_light fire_hydrant stop_sign
parking_meter bench bird cat dog horse sheep cow elephant bear zebra giraffe backpack umbrella handbag tie suitcase
frisbee skis snowboard sports_ball kite baseball_bat baseball_glove skateboard surfboard tennis_racket bottle
wine_glass cup fork knife spoon bowl banana apple sandwich orange broccoli carrot hot_dog pizza donut cake chair
sofa pottedplant bed diningtable toilet tvmonitor laptop mouse remote keyboard cell_phone microwave oven toaster
sink refrigerator book clock vase scissors teddy_bear hair_drier toothbrush""".split()


def write_assets(root):
    """(Re)write the packaged cfgs / anchors / class names under `root` (= yolo_amd/)."""
    os.makedirs(os.path.join(root, "darknet_data"), exist_ok=True)
    os.makedirs(os.path.join(root, "model_data"), exist_ok=True)
    files = {
        "darknet_data/yolov3.cfg": yolov3_cfg(),
        "darknet_data/yolov3-tiny.cfg": yolov3_tiny_cfg(),
        "model_data/yolo_anchors_pre-trained.txt": "".join("%d %d\n" % a for a in COCO_ANCHORS),
        "model_data/yolo_tiny_anchors.txt": "".join("%d %d\n" % a for a in TINY_ANCHORS),
        "model_data/coco_classes.txt": "".join(c.replace("_", " ") + "\n" for c in COCO_CLASSES),
    }
    for rel, text in files.items():
        with open(os.path.join(root, rel), "w") as fh:
            fh.write(text)
    return sorted(files)


if __name__ == "__main__":
    print("\n".join(write_assets(os.path.dirname(os.path.abspath(__file__)))))
