"""yolo_amd: MI355X-native YOLOv3 / YOLOv3-tiny inference behind the Python API of
viplix3/YOLO (build_model.yolo / load_weights, utils.utils.get_head / correct_boxes /
get_boxes_and_scores, inference.predict / run_inference).  All arithmetic runs in
hand-written gfx950 HIP kernels (yolo_amd/csrc, C-ABI in include/yolo3hip.h); torch is
used for device memory, streams and torch.distributed only.  There is no CPU fallback."""
from . import config  # noqa: F401

__all__ = ["config", "build_model", "inference", "postprocess", "planner", "darknet_weights", "darknet_cfg"]
