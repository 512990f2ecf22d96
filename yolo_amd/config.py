"""Configuration values the inference path reads; same attribute names as the
reference's config.py (config.py:2,3,14,15,18-22,35,36,39,44-49).  Training-only
knobs are not carried over.  Paths point at the files packaged with yolo_amd."""
import os

_HERE = os.path.dirname(os.path.abspath(__file__))

input_shape = 320            # config.py:2 (the benchmarks override it with 416 / 608)
max_boxes = 51               # config.py:3   per-class NMS cap
norm_decay = 0.9             # config.py:14  (training-only; accepted, unused)
weight_decay = 0.0005        # config.py:15  (training-only; accepted, unused)
feature_extractor_conv_count = 52   # config.py:18
num_anchors = 9              # config.py:19
num_anchors_per_scale = 3    # config.py:20
num_scales = 3               # config.py:21
num_classes = 80             # config.py:22
score_threshold = 0.6        # config.py:35
nms_threshold = 0.3          # config.py:36
gpu_num = "0"                # config.py:39
# config.py:44 then :48 assigns anchors_path twice (the second, ./anchors.txt, is not shipped);
# the packaged COCO anchors are the first assignment's file.
anchors_path = os.path.join(_HERE, "model_data", "yolo_anchors_pre-trained.txt")
tiny_anchors_path = os.path.join(_HERE, "model_data", "yolo_tiny_anchors.txt")   # from yolov3-tiny.cfg:134
yolov3_cfg_path = os.path.join(_HERE, "darknet_data", "yolov3.cfg")            # config.py:45
yolov3_tiny_cfg_path = os.path.join(_HERE, "darknet_data", "yolov3-tiny.cfg")
yolov3_spp_cfg_path = os.path.join(_HERE, "darknet_data", "yolov3-spp.cfg")   # stem 'yolov3-spp' (build_model.py:70-71); cfg generated here, the reference ships none
yolov3_weights_path = os.path.join(_HERE, "darknet_data", "yolov3.weights")    # config.py:46
classes_path = os.path.join(_HERE, "model_data", "coco_classes.txt")           # config.py:49

compute_dtype = "bf16"       # "bf16" or "fp32" (new: the reference is fp32-only)
