"""Output side of the image-dir contract (reference inference.py:209-215, 284-323): class colours and the
annotated image.  CPU only."""
import colorsys
import random

import numpy as np

from yolo_amd import inference


def test_class_colors_follow_the_reference_recipe():
    n = 80
    cols = inference.class_colors(n)
    # the reference's own steps, with the global generator it uses (inference.py:209-215)
    want = [colorsys.hsv_to_rgb(x / n, 1., 1.) for x in range(n)]
    want = [(int(r * 255), int(g * 255), int(b * 255)) for r, g, b in want]
    random.seed(10101)
    random.shuffle(want)
    random.seed(None)
    assert cols == want and len(set(cols)) > 70


def test_draw_detections_outline_label_and_clipping():
    img = np.full((200, 300, 3), 255, dtype=np.uint8)
    names = ["a", "b", "c"]
    cols = inference.class_colors(3)
    boxes = np.array([[50.2, 60.4, 120.6, 200.5], [-10.0, 250.0, 40.0, 400.0]], dtype=np.float32)   # second: clipped to the image
    out = np.asarray(inference.draw_detections(img, boxes, np.array([0.9, 0.7], dtype=np.float32),
                                               np.array([1, 2], dtype=np.int32), names, cols))
    assert out.shape == img.shape
    t = (200 + 300) // 500                                   # thickness 1
    assert t == 1
    top, left, bottom, right = 50, 60, 121, 201              # floor(v + 0.5), inference.py:302-306
    assert tuple(out[bottom, 100]) == cols[1] and tuple(out[100, left]) == cols[1] and tuple(out[100, right]) == cols[1]
    assert tuple(out[100, 130]) == (255, 255, 255)           # interior untouched
    # the label sits above the box when there is room: a filled patch of the class colour over the top-left corner
    assert tuple(out[top - 2, left + 1]) in (cols[1], (0, 0, 0)) or (out[top - 8:top, left:left + 30] == np.array(cols[1])).all(axis=-1).any()
    # second box is clipped: left edge at x = 250 drawn from y = 0 down to 40, right edge at the image border
    assert tuple(out[30, 250]) == cols[2]
    lines = inference.label_lines(boxes, np.array([0.9, 0.7], dtype=np.float32), np.array([1, 2]), (200, 300))
    assert lines[0].startswith("250,0,300,40,2,") and lines[1].startswith("60,50,201,121,1,")


def test_frozen_graph_node_names():
    """Terminal node names of the reference's frozen graph (convert_to_pb.py:11; predict_from_pb.py:221-225)."""
    import pytest
    from yolo_amd import frozen_graph
    from yolo_amd.build_model import Network
    assert frozen_graph.terminal_node_names == ["input_image", "input_shape", "scale_1", "scale_2", "scale_3"]
    net = Network.__new__(Network)          # no device needed to look names up
    net.out_index = [0, 1]                  # a two-head (tiny) network has no scale_3
    g = frozen_graph.FrozenGraph(net)
    assert g.get_tensor_by_name("scale_2:0").name == "scale_2"
    assert g.get_tensor_by_name("input_image:0") == g.get_tensor_by_name("input_image")
    with pytest.raises(KeyError):
        g.get_tensor_by_name("scale_3:0")
    with pytest.raises(KeyError):
        g.get_tensor_by_name("scale_1:1")
    with pytest.raises(ValueError):
        g.run("scale_1:0", {})
    assert list(g.run("input_shape:0", {"input_shape:0": (480, 640)})) == [480, 640]
