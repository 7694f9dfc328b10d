"""bf16 engine with an f32 warm-up, for the norm-heavy detector classes (FCOS, CenterNet).

At RANDOM INITIALISATION the bf16 engine's gradients of these classes lose their direction towards the input (identity-free conv + norm stacks
re-amplify the 2^-9 rounding of every stored activation: filter-gradient cosine against the f32 engine 0.3-0.4 on the input-side third of the layers,
DESIGN.md 3h / 3i / 5).  After a few hundred optimizer steps that is over: measured on synthetic data after 300 f32 steps (tools/bf16_after_training.py,
profiles/r03i_bf16_after_training.md) -- CenterNet 0.30 -> 0.97 (minimum over all layers 0.96), FCOS 0.41 -> 0.94 (minimum 0.88) -- and with pre-trained
weights it never applies.  So the classes default to the bf16 engine (3.6-4.4x the f32 engine's step rate) and, when they start from random initialisation
with no `compute_dtype` given, run their first `f32_warmup_steps` (default 300) optimizer steps on a TWIN instance built on the f32 engine, then move
parameters, optimizer state and moving statistics over and free the twin.  An explicit `compute_dtype` ('bf16' | 'f32') is taken literally (no warm-up unless
`f32_warmup_steps` is given too); loading weights (load_weight / load_pretrained_weight / load_oracle_params) cancels a pending warm-up.

The gate: tests/test_gpu_bf16_gate.py trains each class 300 steps in f32 and requires the cosines above; RetinaNet does not pass it (0.71 after 300 steps) and
keeps the f32 engine."""
import torch

F32_WARMUP_DEFAULT = 300


class F32Warmup:
    OPT_BUFFERS = ('Mom',)            # flat optimizer-state buffers laid out like P (CenterNet: ('M1', 'M2'))

    def _warmup_setup(self, config, data_provider, explicit_dtype):
        from ._lib import BF16
        self._twin = None
        self._twin_args = (dict(config), data_provider)
        self._last_batch = None
        steps = config.get('f32_warmup_steps', None)
        if steps is None:
            steps = F32_WARMUP_DEFAULT if (not explicit_dtype and self.dev.type == 'cuda') else 0
        self.f32_warmup_steps = int(steps) if (self.mode == 'train' and self.DT == BF16) else 0

    def _warming(self):
        return self.f32_warmup_steps > self.global_step

    def cancel_warmup(self):
        """weights were loaded: the run does not start from random initialisation"""
        if self._twin is not None:
            self._twin = None
            torch.cuda.empty_cache()
        self.f32_warmup_steps = 0

    def _sync_from_twin(self):
        """mid-warm-up the live weights are the twin's: whoever reads this instance's buffers (export_params, export_tf_variables, save_weight) syncs first"""
        if getattr(self, '_twin', None) is not None:
            self._copy_state(self._twin, self)

    @staticmethod
    def _copy_state(src, dst):
        # the flat buffers of the two engines differ in the input-channel padding of the filters (16-byte chunks: 8 bf16 | 4 f32): copy variable by variable
        for k in src.pinfo:
            for name in (None,) + tuple(src.OPT_BUFFERS):
                v = src.param(k, None if name is None else getattr(src, name))
                d = dst.param(k, None if name is None else getattr(dst, name))
                if k.endswith('.w'):
                    c = min(v.shape[-1], d.shape[-1])
                    d.zero_()
                    d[..., :c].copy_(v[..., :c])
                else:
                    d.copy_(v.view(d.shape))
        if hasattr(src, 'S') and isinstance(src.S, torch.Tensor):
            dst.S.copy_(src.S.to(dst.dev))
        dst.global_step = src.global_step
        dst._refresh_operand_copies()

    def _twin_get(self):
        if self._twin is None:
            cfg, prov = self._twin_args
            # (round 4: the twin's convolutions run as split bf16 products -- f32 tensors, ODTK_F32X3 descriptors: every filter gradient within cosine 0.998 of the exact
            #  f32 engine's at random initialisation for all these classes (profiles/r04x_gate_f32x3_all_classes.md), 1.7-2.2x its throughput)
            self._twin = type(self)(dict(cfg, compute_dtype='f32x3', f32_warmup_steps=0), prov)
            self._copy_state(self, self._twin)
            if self.dist is not None:
                self._twin.attach_data_parallel(self.dist.red.group)
            if self._last_batch is not None:
                self._twin._set_batch_engine(*self._last_batch)
        return self._twin

    def save_weight(self, mode, path):
        self._sync_from_twin()                                   # mid-warm-up: the live weights are the twin's
        return self._save_weight_engine(mode, path)

    def set_batch(self, images, ground_truth):
        self._last_batch = (images, ground_truth)
        if self._warming():
            self._twin_get()._set_batch_engine(images, ground_truth)
        else:
            self._set_batch_engine(images, ground_truth)

    def train_step(self, lr):
        if not self._warming():
            return self._train_step_engine(lr)
        t = self._twin_get()
        loss = t._train_step_engine(lr)
        self.global_step = t.global_step
        if not self._warming():                                  # that was the last warm-up step: the bf16 engine takes over from here
            self._copy_state(t, self)
            self._twin = None
            if self._last_batch is not None:
                self._set_batch_engine(*self._last_batch)
            if self.dev.type == 'cuda':
                torch.cuda.empty_cache()
            if self.verbose:
                print(f'[odtk] {type(self).__name__}: f32 warm-up of {self.f32_warmup_steps} steps done, continuing on the bf16 engine')
        return loss
