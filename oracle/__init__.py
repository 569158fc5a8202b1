"""CPU restatement of the reference's detection hot path -- TEST INFRASTRUCTURE ONLY.

Nothing under this package is part of the product: only tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py import it, and only as the checker (or as the timed CPU baseline), never
as the thing shipped or measured.  Each function cites the reference lines it follows
(paths relative to the reference root, he13689/yolo).

Pinning status
  * decode (both families), the family-A numpy NMS / postprocess, the candidate stage of
    non_max_suppression, the conv blocks / whole networks and the metric helpers are pinned by golden
    vectors generated from the imported reference (tools/gen_golden.py -> tests/golden/*.npz).
  * torchvision.ops.nms (called at yolov5_official/utils.py:182) is NOT part of the reference tree, no
    version of it is pinned there and torchvision is not installed: `torchvision_nms` below restates
    its documented algorithm and is "parity unpinned" for that greedy core.
"""
