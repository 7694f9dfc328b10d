"""Per-class threshold + NMS tail shared by the inference branches of RetinaNet / YOLOv3 / FCOS (TEST INFRASTRUCTURE ONLY).

Restates the loop `for i in range(classes): boolean_mask by score >= thr; tf.image.non_max_suppression; gather; concat`
(RetinaNet.py:239-256, YOLOv3.py:351-368, FCOS.py:248-265) on top of the NonMaxSuppressionV3 restatement of
oracle/ssd300_ref.py.  Pinned by the reference's own loops run on the shim (tests/golden/*det*)."""
import numpy as np
import torch

from . import ssd300_ref as R


def per_class_nms(conf, boxes, num_classes, score_thr, max_boxes, iou_thr, row_mask=None):
    """conf [L, >= num_classes], boxes [L, 4]; row_mask [L] bool: rows removed before the per-class filter (RetinaNet's
    background-arg-max mask).  Returns scores [K], bbox [K, 4], class_id [K] int32 in the reference's concat order."""
    if row_mask is not None:
        conf, boxes = conf[row_mask], boxes[row_mask]
    s_out, b_out, c_out = [], [], []
    for c in range(num_classes):
        m = conf[:, c] >= score_thr
        sc, bx = conf[m, c], boxes[m]
        sel = torch.from_numpy(R.nms(bx.numpy(), sc.numpy(), max_boxes, iou_thr).astype(np.int64))
        s_out.append(sc[sel]); b_out.append(bx[sel])
        c_out.append(torch.full((sel.shape[0],), c, dtype=torch.int32))
    return torch.cat(s_out), torch.cat(b_out, 0).reshape(-1, 4), torch.cat(c_out)
