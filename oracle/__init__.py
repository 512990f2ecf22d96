"""CPU oracle for the YOLOv3 inference hot path.

TEST INFRASTRUCTURE ONLY.  This package is a numpy / plain-C restatement of the
reference's algorithm (viplix3/YOLO: build_model.py, utils/utils.py,
inference.py).  It is the *checker* for the HIP path; it is never part of the
product path.  Only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline`
leg of `bench.py` may import it.

PARITY UNPINNED: the reference holds no tests, golden vectors or fixtures for
this path, and its arithmetic lives in tensorflow-gpu==1.12.0 /
opencv-python==3.4.4.19 (requirements.txt:14,23), neither of which is
installable here (no network).  The restatement is instead pinned by
  (1) hand-derived known-answer vectors (tests/golden/*.json),
  (2) op-by-op cross-checks against torch-CPU (an independent implementation),
  (3) Darknet's published FLOP counts reproduced from the cfgs.
"""
