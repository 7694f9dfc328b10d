"""PFPNetR (VGG-16 to conv4_3 + parallel feature pyramid + RefineDet's ARM / TCB / ODM) behind the reference's class surface, on libodtk.

Reference: /root/reference/PFPNetR.py (class PFPNetR; `input_size` a multiple of 64, 320 in testpfpnetr.py)
  * constructor, config keys ............. :11-53   (as RefineDet plus nothing: the same keys)
  * VGG trunk conv1_1 .. conv4_3 ......... :231-313  (conv + bias + ReLU, three 2x2 / s2 pools) -> fh1, stride 8
  * fh2, fh3, fh4 ........................ :315-324  tf.image.resize_bilinear(fh1, 1/2, 1/4, 1/8, align_corners=True)  (odtk_resize_bilinear2_*)
  * 85-channel pyramid branches .......... :330-364  1x1 + BN + ReLU of every fh_k; up-path blocks 4x4 / s2 transposed conv + BN, + the level below's branch,
                                                      1x1 + BN + ReLU; down-path blocks 2x2 average pool, 1x1 + BN
  * features ............................. :366-396, :76-94   tf.concat of the four level-k tensors (512 + 3 * 85 = 767 channels, pieces at channel offsets
                                                      that are not 16-byte aligned: odtk_copy_channels), feat1 / feat2 L2-normalised
  * ARM / TCB / ODM, loss, inference ..... the text of RefineDet.py (same functions, diffed): refinedet.RefineDet320's
Everything but the feature extractor is inherited; the graph engine (per-activation gradient buffers, write-or-accumulate decided at build time,
the ReLU mask of a bias + ReLU activation applied by each of its consumers) is refinedet.py's with the 'resize', 'avgpool', 'add' and 'concat' kinds.
"""
from __future__ import annotations

from .refinedet import NA, RefineDet320

VGG_SEQ = [("conv1_1", 3, 64), ("conv1_2", 64, 64), "pool1", ("conv2_1", 64, 128), ("conv2_2", 128, 128), "pool2",
           ("conv3_1", 128, 256), ("conv3_2", 256, 256), ("conv3_3", 256, 256), "pool3",
           ("conv4_1", 256, 512), ("conv4_2", 512, 512), ("conv4_3", 512, 512)]
CB = 512 // 6                                                   # 85 channels per branch (PFPNetR.py:330)
UP = [(2, 1), (3, 2), (3, 1), (4, 3), (4, 2), (4, 1)]          # creation order of the up-path blocks fl<a>_<b>
DOWN = [(1, 2), (1, 3), (1, 4), (2, 3), (2, 4), (3, 4)]
FEAT_C = 512 + 3 * CB


def layer_specs(num_classes):
    """[(name, kind, cin, cout, k, stride, dil, relu)] in TensorFlow's creation order (oracle/pfpnet_net_ref.layer_specs' names)"""
    s = []
    for l in VGG_SEQ:
        if isinstance(l, tuple):
            s.append((l[0], 'vgg', l[1], l[2], 3, 1, 1, True))
    for k in range(1, 5):
        s.append((f'fl{k}', 'conv', 512, CB, 1, 1, 1, True))
    for a, b in UP:
        s.append((f'fl{a}_{b}d', 'dconv', CB, CB, 4, 2, 1, False))
        s.append((f'fl{a}_{b}c', 'conv', CB, CB, 1, 1, 1, True))
    for a, b in DOWN:
        s.append((f'fl{a}_{b}', 'conv', CB, CB, 1, 1, 1, False))

    def head(prefix, cin, ncls):
        c = cin
        for j in range(1, 5):
            s.append((f'{prefix}.c{j}', 'conv', c, 256, 3, 1, 1, True)); c = 256
        s.append((f'{prefix}.loc', 'conv', 256, 4 * NA, 3, 1, 1, False))
        s.append((f'{prefix}.conf', 'conv', 256, ncls * NA, 3, 1, 1, False))
    for l in range(4):
        head(f'arm{l + 1}', FEAT_C, 2)
    for l in (4, 3, 2, 1):
        s.append((f'tcb{l}.c1', 'conv', FEAT_C, 256, 3, 1, 1, True))
        s.append((f'tcb{l}.c2', 'conv', 256, 256, 3, 1, 1, l == 4))
        if l < 4:
            s.append((f'tcb{l}.d', 'dconv', 256, 256, 4, 2, 1, False))
    for l in range(4):
        head(f'odm{l + 1}', 256, num_classes)
    return s


class PFPNetR(RefineDet320):
    VGG_SEQ = VGG_SEQ
    L2_AFTER = 'fl3_4'                      # the two L2-norm scalars are created after the feature extractor's last layer (PFPNetR.py:79, :81)
    NAME = 'PFPNetR'

    @staticmethod
    def layer_specs(num_classes):
        return layer_specs(num_classes)

    def _build_features(self, h):
        """PFPNetR.py:230-401 -> feat1 .. feat4 (feat1, feat2 L2-normalised, :76-94)"""
        x = self.input
        for l in self.VGG_SEQ:
            x = h.vgg(l[0], x) if isinstance(l, tuple) else h.pool(l, x, 2, 2)
        fh = {1: x}
        for k in (2, 3, 4):
            fh[k] = h.resize(f'fh{k}', x, x.H >> (k - 1), x.W >> (k - 1))
        fl = {(k, k): h.bn(f'fl{k}', fh[k]) for k in range(1, 5)}
        for a, b in UP:                                         # fl<a>_<b> = relu(bn(1x1(bn(dconv(fl<a>_<b+1>)) + fl<b>)))
            d = h.bn(f'fl{a}_{b}d', fl[(a, b + 1)])
            fl[(a, b)] = h.bn(f'fl{a}_{b}c', h.add(f'fl{a}_{b}s', d, fl[(b, b)]))
        for a, b in DOWN:                                       # fl<a>_<b> = bn(1x1(avg_pool(fl<a>_<b-1>)))
            fl[(a, b)] = h.bn(f'fl{a}_{b}', h.avgpool(f'fl{a}_{b}p', fl[(a, b - 1)]))
        cat = [h.concat(f'cat{k}', [fh[k] if a == k else fl[(a, k)] for a in range(1, 5)]) for k in range(1, 5)]
        return [h.l2norm('feat1', cat[0], 'feat1_l2_norm'), h.l2norm('feat2', cat[1], 'feat2_l2_norm'), cat[2], cat[3]]
