"""Host-side owner of one HIP engine context (one per process / GPU).

PyTorch is used here for what it is good at -- device memory, streams, the
``torch.distributed`` collective on the flat gradient buffer -- while every
FLOP of the fine-tune step runs in ``libpevit_hip.so`` (include/pevit_hip.h).
The trainable parameters live in ONE flat f32 buffer in the reference's
``named_parameters()`` order (adapters, then ``layers.0.weight/bias`` of the
Classifier head, kadaptation_clip.py:132); ``param_views()`` exposes them under
the reference's names so that state-dicts, the requires_grad-by-substring rule
and DP all-reduce see exactly the tensors the reference has.
"""
from __future__ import annotations

import ctypes as C
from collections import OrderedDict

import torch

from . import _lib, dp
from .synth import VitArch

PHM_DIM_KADAPT = 32     # model.py:485
BOTTLENECK = 64         # adapter_model.py:305, compacter_model.py:472
PHM_DIM_COMPACTER = 4   # compacter_model.py:512


def adapter_param_spec(method: str, width: int, layers: int, lora_rank: int = 4):
    """(name, shape, trainable) of the tensors the reference adds to the OpenAI layout, in the
    reference's ``named_parameters()`` order (verified against the golden
    fixtures' name lists in tests/test_host_api.py)."""
    E, t = width, "visual.transformer."
    out = []
    if method == "kadaptation":
        n = PHM_DIM_KADAPT
        out += [(t + "phm_rule1_left", (n, n, 1), True), (t + "phm_rule1_right", (n, 1, n), True),
                (t + "phm_rule2_left", (n, n, 1), True), (t + "phm_rule2_right", (n, 1, n), True)]
        for i in range(layers):
            a = f"{t}resblocks.{i}.attn."
            out += [(a + "q_proj_adapter1_left", (n, E // n, 1), True), (a + "q_proj_adapter1_right", (n, 1, E // n), True),
                    (a + "v_proj_adapter1_left", (n, E // n, 1), True), (a + "v_proj_adapter1_right", (n, 1, E // n), True),
                    (a + "b", (E,), True)]
    elif method == "lora":
        for i in range(layers):
            a = f"{t}resblocks.{i}.attn."
            out += [(a + "q_proj_adapter1.weight", (lora_rank, E), True), (a + "q_proj_adapter2.weight", (E, lora_rank), True),
                    (a + "v_proj_adapter1.weight", (lora_rank, E), True), (a + "v_proj_adapter2.weight", (E, lora_rank), True)]
    elif method == "adapter":
        for i in range(layers):
            a = f"{t}resblocks.{i}.adapter."
            out += [(a + "adapter_norm_before.weight", (E,), True), (a + "adapter_norm_before.bias", (E,), True),
                    (a + "adapter_down.1.weight", (BOTTLENECK, E), True), (a + "adapter_down.1.bias", (BOTTLENECK,), True),
                    (a + "adapter_up.weight", (E, BOTTLENECK), True), (a + "adapter_up.bias", (E,), True)]
    elif method == "compacter":
        n = PHM_DIM_COMPACTER
        out += [(t + "phm_rule", (n, n, n), False)]          # name lacks 'compacter' -> frozen (compacter_clip.py:122)
        for i in range(layers):
            a = f"{t}resblocks.{i}.compacter."
            out += [(a + "adapter_norm_before.weight", (E,), True), (a + "adapter_norm_before.bias", (E,), True),
                    (a + "adapter_down.1.W_left", (n, E // n, 1), True), (a + "adapter_down.1.W_right", (n, 1, BOTTLENECK // n), True),
                    (a + "adapter_down.1.b", (BOTTLENECK,), True),
                    (a + "adapter_up.W_left", (n, BOTTLENECK // n, 1), True), (a + "adapter_up.W_right", (n, 1, E // n), True),
                    (a + "adapter_up.b", (E,), True)]
    elif method == "none":
        pass
    else:
        raise ValueError(f"unknown PEFT method {method!r}")
    return out


def _numel(shape):
    n = 1
    for s in shape:
        n *= s
    return n


class HipEngine:
    def __init__(self, arch: VitArch, method: str, num_classes: int, max_batch: int, lora_rank: int = 4,
                 device: str | torch.device = "cuda:0", weight_format: str = "bf16"):
        if not torch.cuda.is_available():
            raise _lib.PevitError("HipEngine needs a ROCm GPU (gfx950); there is no CPU fallback")
        self.lib = _lib.load()
        self.arch, self.method, self.num_classes = arch, method, num_classes
        self.lora_rank, self.max_batch = lora_rank, max_batch
        if weight_format not in _lib.WEIGHT_FORMATS:
            raise ValueError(f"weight_format {weight_format!r}: expected one of {sorted(_lib.WEIGHT_FORMATS)}")
        self.weight_format = weight_format
        self.device = torch.device(device)
        torch.cuda.set_device(self.device)
        dims = _lib.PevitDims(arch.width, arch.layers, arch.patch, arch.resolution, arch.embed_dim,
                              _lib.METHOD_IDS[method], lora_rank, num_classes, _lib.WEIGHT_FORMATS[weight_format])
        self._ctx = C.c_void_p()
        _lib.check(self.lib.pevit_ctx_create(C.byref(dims), C.byref(self._ctx)), "pevit_ctx_create")
        self.arena = torch.zeros(self.lib.pevit_arena_bytes(self._ctx), dtype=torch.uint8, device=self.device)
        self.workspace = torch.empty(self.lib.pevit_workspace_bytes(self._ctx, max_batch), dtype=torch.uint8,
                                     device=self.device)
        _lib.check(self.lib.pevit_bind(self._ctx, _lib.ptr(self.arena), self.arena.numel(), _lib.ptr(self.workspace),
                                       self.workspace.numel(), max_batch), "pevit_bind")
        self.n_tower = self.lib.pevit_num_tower_params(self._ctx)
        self.n_params = self.lib.pevit_num_params(self._ctx)
        self.params = torch.zeros(self.n_params, dtype=torch.float32, device=self.device)
        self.grads = torch.zeros_like(self.params)
        self.momentum = torch.zeros_like(self.params)
        mask = (C.c_ubyte * self.n_params)()
        _lib.check(self.lib.pevit_param_grad_mask(self._ctx, mask, self.n_params), "pevit_param_grad_mask")
        self.grad_mask_host = torch.frombuffer(bytearray(mask), dtype=torch.uint8).clone()
        self.grad_mask = self.grad_mask_host.to(self.device)
        _lib.check(self.lib.pevit_set_params(self._ctx, _lib.ptr(self.params), _lib.ptr(self.grads),
                                             _lib.ptr(self.momentum), _lib.ptr(self.grad_mask)), "pevit_set_params")
        # BatchNorm1d(D, affine=False) buffers of the Classifier (kadaptation_clip.py:128-131)
        self.running_mean = torch.zeros(arch.embed_dim, dtype=torch.float32, device=self.device)
        self.running_var = torch.ones(arch.embed_dim, dtype=torch.float32, device=self.device)
        self._spec = [(n, s) for n, s, tr in adapter_param_spec(method, arch.width, arch.layers, lora_rank) if tr]
        total = sum(_numel(s) for _, s in self._spec)
        if total != self.n_tower:
            raise _lib.PevitError(f"host/engine parameter layout mismatch: {total} vs {self.n_tower}")
        self._steps = 0
        self.forward_generation = 0      # bumped by EVERY forward: each one overwrites the single activation workspace
        self.block_generation = {}       # per block: bumped by every forward THROUGH that block (blocks keep their own activations)
        self._logits = torch.empty((max_batch, num_classes), dtype=torch.float32, device=self.device)
        self._loss = torch.zeros(1, dtype=torch.float32, device=self.device)

    def __del__(self):
        try:
            if getattr(self, "_ctx", None):
                self.lib.pevit_ctx_destroy(self._ctx)
                self._ctx = None
        except Exception:
            pass

    def tune(self, key: str, value: int) -> int:
        """A/B-measurement knob of THIS context (include/pevit_hip.h: pevit_tune)."""
        return self.lib.pevit_tune(self._ctx, key.encode(), int(value))

    def ensure_batch(self, batch: int):
        """Grow the activation workspace so that steps of ``batch`` images fit (re-binds; weights stay)."""
        if batch <= self.max_batch:
            return
        torch.cuda.synchronize(self.device)
        self.workspace = None
        self.workspace = torch.empty(self.lib.pevit_workspace_bytes(self._ctx, batch), dtype=torch.uint8,
                                     device=self.device)
        _lib.check(self.lib.pevit_bind(self._ctx, _lib.ptr(self.arena), self.arena.numel(), _lib.ptr(self.workspace),
                                       self.workspace.numel(), batch), "pevit_bind")
        self._logits = torch.empty((batch, self.num_classes), dtype=torch.float32, device=self.device)
        self.max_batch = batch

    # ------------------------------------------------------------------ parameters
    def param_views(self, buf: torch.Tensor | None = None) -> "OrderedDict[str, torch.Tensor]":
        """Views of the flat buffer under the reference's parameter names (+ head).  (Pipelined DP: the current stream first
        waits for the update still running on the second stream -- a checkpoint / state export right behind a step must not read
        half-updated parameters.)"""
        self.dp_flush()
        buf = self.params if buf is None else buf
        out, off = OrderedDict(), 0
        for name, shape in self._spec:
            n = _numel(shape)
            out[name] = buf[off:off + n].view(shape)
            off += n
        D, Cc = self.arch.embed_dim, self.num_classes
        out["layers.0.weight"] = buf[off:off + Cc * D].view(Cc, D); off += Cc * D
        out["layers.0.bias"] = buf[off:off + Cc].view(Cc); off += Cc
        assert off == self.n_params
        return out

    def grad_views(self):
        self.dp_flush()
        return self.param_views(self.grads)

    # ------------------------------------------------------------------ frozen weights
    def load_state_dict(self, sd, prefix: str = "visual."):
        """Frozen backbone from an OpenAI-layout state-dict (any float dtype, any device)."""
        self.dp_flush()
        s = _lib.stream_ptr()

        def dev(key):
            return sd[prefix + key].detach().to(device=self.device, dtype=torch.float32).contiguous()

        keep = []
        for i in range(self.arch.layers):
            b = f"transformer.resblocks.{i}."
            ts = [dev(b + k) for k in ("attn.in_proj_weight", "attn.in_proj_bias", "attn.out_proj.weight",
                                       "attn.out_proj.bias", "ln_1.weight", "ln_1.bias", "mlp.c_fc.weight",
                                       "mlp.c_fc.bias", "mlp.c_proj.weight", "mlp.c_proj.bias", "ln_2.weight", "ln_2.bias")]
            keep.append(ts)
            _lib.check(self.lib.pevit_load_block(self._ctx, s, i, *[_lib.ptr(t) for t in ts]), "pevit_load_block")
        ts = [dev(k) for k in ("conv1.weight", "class_embedding", "positional_embedding", "ln_pre.weight",
                               "ln_pre.bias", "ln_post.weight", "ln_post.bias", "proj")]
        keep.append(ts)
        _lib.check(self.lib.pevit_load_stem(self._ctx, s, *[_lib.ptr(t) for t in ts]), "pevit_load_stem")
        if self.method == "compacter":
            r = dev("transformer.phm_rule"); keep.append([r])
            _lib.check(self.lib.pevit_load_phm_rule(self._ctx, s, _lib.ptr(r)), "pevit_load_phm_rule")
        torch.cuda.current_stream().synchronize()     # the temporaries above may now be freed
        del keep
        self.load_trainable(sd)

    def load_phm_rule(self, rule: torch.Tensor):
        """Compacter's shared, frozen (4,4,4) rule (compacter_model.py:511-519) into the arena."""
        r = rule.detach().to(device=self.device, dtype=torch.float32).contiguous()
        _lib.check(self.lib.pevit_load_phm_rule(self._ctx, _lib.stream_ptr(), _lib.ptr(r)), "pevit_load_phm_rule")
        torch.cuda.current_stream().synchronize()

    def load_trainable(self, sd):
        """Adapter / head tensors present in ``sd`` overlay the current values (model.py:1247-1250)."""
        views = self.param_views()
        with torch.no_grad():
            for name, v in views.items():
                if name in sd:
                    v.copy_(sd[name].to(device=self.device, dtype=torch.float32).view_as(v))

    # ------------------------------------------------------------------ hot path
    def set_input_normalization(self, mean, std):
        """Preprocessing of uint8 batches, x = (u8 / 255 - mean[c]) / std[c] (the reference's ToTensor + Normalize,
        feature.py:537-542 with INPUT.MEAN / INPUT.STD), applied inside the patch gather of the engine."""
        m = (C.c_float * 3)(*[float(v) for v in mean]); sd = (C.c_float * 3)(*[float(v) for v in std])
        _lib.check(self.lib.pevit_set_input_norm(self._ctx, m, sd), "pevit_set_input_norm")
        self._input_norm = (tuple(float(v) for v in mean), tuple(float(v) for v in std))

    def _check_batch(self, images, labels=None):
        """The C ABI takes raw pointers: reject anything that is not what it will read (f32 NCHW images -- or uint8 pixels once
        set_input_normalization has been called -- and int64 class indices, contiguous, on this engine's device)."""
        a = self.arch
        ok_dtype = images.dtype == torch.float32 or (images.dtype == torch.uint8 and getattr(self, "_input_norm", None) is not None)
        if images.device != self.device or not ok_dtype or not images.is_contiguous() or \
                tuple(images.shape[1:]) != (3, a.resolution, a.resolution):
            raise _lib.PevitError(f"images must be a contiguous float32 tensor (B,3,{a.resolution},{a.resolution}) on {self.device} "
                                  f"(or uint8 after set_input_normalization); got {tuple(images.shape)} {images.dtype} on {images.device}")
        if images.shape[0] > self.max_batch:
            raise _lib.PevitError(f"batch {images.shape[0]} exceeds the bound workspace ({self.max_batch}): call ensure_batch")
        if labels is not None and (labels.device != self.device or labels.dtype != torch.int64 or not labels.is_contiguous()
                                   or tuple(labels.shape) != (images.shape[0],)):
            raise _lib.PevitError(f"labels must be a contiguous int64 tensor ({images.shape[0]},) on {self.device}; "
                                  f"got {tuple(labels.shape)} {labels.dtype} on {labels.device}")

    def transformer_forward(self, x_nbe: torch.Tensor, save: bool = True) -> torch.Tensor:
        self.dp_flush()
        N, B, E = x_nbe.shape
        if (N, E) != (self.arch.tokens, self.arch.width) or x_nbe.device != self.device:
            raise _lib.PevitError(f"transformer_forward expects ({self.arch.tokens}, B, {self.arch.width}) on {self.device}, "
                                  f"got {tuple(x_nbe.shape)} on {x_nbe.device}")
        self.forward_generation += 1
        for l in range(self.arch.layers):
            self.block_generation[l] = self.block_generation.get(l, 0) + 1
        x = x_nbe.contiguous().float()
        y = torch.empty_like(x)
        _lib.check(self.lib.pevit_transformer_forward(self._ctx, _lib.stream_ptr(), _lib.ptr(x), _lib.ptr(y), B, int(save)),
                   "pevit_transformer_forward")
        return y

    def blocks_forward(self, x_nbe: torch.Tensor, l_lo: int, l_hi: int, save: bool = True) -> torch.Tensor:
        """Blocks [l_lo, l_hi) on (N,B,E) activations (the reference's ResidualAttentionBlock.forward for one block)."""
        self.dp_flush()
        N, B, E = x_nbe.shape
        if (N, E) != (self.arch.tokens, self.arch.width) or x_nbe.device != self.device:
            raise _lib.PevitError(f"blocks_forward expects ({self.arch.tokens}, B, {self.arch.width}) on {self.device}")
        # every forward -- also one through a few blocks -- overwrites activations a pending whole-tower backward would read;
        # the output of block l_hi - 1 also lands in the saved INPUT of block l_hi
        self.forward_generation += 1
        for l in range(l_lo, min(l_hi + 1, self.arch.layers)):
            self.block_generation[l] = self.block_generation.get(l, 0) + 1
        x = x_nbe.contiguous().float()
        y = torch.empty_like(x)
        _lib.check(self.lib.pevit_blocks_forward(self._ctx, _lib.stream_ptr(), _lib.ptr(x), _lib.ptr(y), B, int(save), l_lo, l_hi),
                   "pevit_blocks_forward")
        return y

    def blocks_backward(self, dy_nbe: torch.Tensor, l_lo: int, l_hi: int, need_dx: bool = True):
        self.dp_flush()
        N, B, E = dy_nbe.shape
        dy = dy_nbe.contiguous().float()
        dx = torch.empty_like(dy) if need_dx else None
        _lib.check(self.lib.pevit_blocks_backward(self._ctx, _lib.stream_ptr(), _lib.ptr(dy), _lib.ptr(dx), B, l_lo, l_hi),
                   "pevit_blocks_backward")
        return dx

    def transformer_backward(self, dy_nbe: torch.Tensor, need_dx: bool = True):
        self.dp_flush()
        N, B, E = dy_nbe.shape
        dy = dy_nbe.contiguous().float()
        dx = torch.empty_like(dy) if need_dx else None
        _lib.check(self.lib.pevit_transformer_backward(self._ctx, _lib.stream_ptr(), _lib.ptr(dy), _lib.ptr(dx), B),
                   "pevit_transformer_backward")
        return dx

    def visual_forward(self, images: torch.Tensor, save: bool = True) -> torch.Tensor:
        self.dp_flush()
        B = images.shape[0]
        img = images.contiguous() if images.dtype == torch.uint8 else images.contiguous().float()
        self._check_batch(img)
        self.forward_generation += 1
        for l in range(self.arch.layers):
            self.block_generation[l] = self.block_generation.get(l, 0) + 1
        feat = torch.empty((B, self.arch.embed_dim), dtype=torch.float32, device=self.device)
        fn = self.lib.pevit_visual_forward_u8 if img.dtype == torch.uint8 else self.lib.pevit_visual_forward
        _lib.check(fn(self._ctx, _lib.stream_ptr(), _lib.ptr(img), _lib.ptr(feat), B, int(save)), "pevit_visual_forward")
        return feat

    def visual_backward(self, dfeat: torch.Tensor):
        self.dp_flush()
        B = dfeat.shape[0]
        d = dfeat.contiguous().float()
        _lib.check(self.lib.pevit_visual_backward(self._ctx, _lib.stream_ptr(), _lib.ptr(d), B), "pevit_visual_backward")

    def head_forward_backward(self, feat, labels, bn_training=True, need_dfeat=True):
        self.dp_flush()
        B = feat.shape[0]
        logits = torch.empty((B, self.num_classes), dtype=torch.float32, device=self.device)
        loss = torch.zeros(1, dtype=torch.float32, device=self.device)
        dfeat = torch.empty_like(feat) if (need_dfeat and labels is not None) else None
        _lib.check(self.lib.pevit_head_forward_backward(
            self._ctx, _lib.stream_ptr(), _lib.ptr(feat.contiguous()), _lib.ptr(labels), _lib.ptr(self.running_mean),
            _lib.ptr(self.running_var), int(bn_training), _lib.ptr(logits), _lib.ptr(loss), _lib.ptr(dfeat), B),
            "pevit_head_forward_backward")
        return logits, loss, dfeat

    def zero_grad(self):
        self.dp_flush()
        _lib.check(self.lib.pevit_zero_grads(self._ctx, _lib.stream_ptr()), "pevit_zero_grads")

    def forward_backward(self, images, labels, bn_training=True, logits_out=None, loss_out=None):
        """zero_grad -> forward -> CE -> backward; gradients land in ``self.grads``.  Returns
        (logits, loss) as device tensors without synchronising (the reference's
        ``loss.item()`` per step, kadaptation_clip.py:354, is the caller's choice).  ``logits_out`` (B, C) / ``loss_out`` (1,):
        caller-owned f32 destinations (an epoch buffer of the harness) instead of the engine's per-step buffers, which the
        next step overwrites."""
        B = images.shape[0]
        self._check_batch(images, labels)
        self.forward_generation += 1
        logits = self._logits[:B] if logits_out is None else self._out_buf(logits_out, (B, self.num_classes))
        loss = self._loss if loss_out is None else self._out_buf(loss_out, (1,))
        fn = self.lib.pevit_train_forward_backward_u8 if images.dtype == torch.uint8 else self.lib.pevit_train_forward_backward
        _lib.check(fn(self._ctx, _lib.stream_ptr(), _lib.ptr(images), _lib.ptr(labels), _lib.ptr(self.running_mean),
                      _lib.ptr(self.running_var), int(bn_training), _lib.ptr(logits), _lib.ptr(loss), B),
                   "pevit_train_forward_backward")
        return logits, loss

    def _out_buf(self, t, shape):
        if t.device != self.device or t.dtype != torch.float32 or not t.is_contiguous() or tuple(t.shape) != tuple(shape):
            raise _lib.PevitError(f"output buffer must be a contiguous float32 tensor {tuple(shape)} on {self.device}; got "
                                  f"{tuple(t.shape)} {t.dtype} on {t.device}")
        return t

    def sgd_step(self, lr, momentum=0.9, weight_decay=0.0, grad_scale=1.0, nesterov=False, _pipelined=False):
        if not _pipelined:             # a direct call while a pipelined update is still running on the second stream
            self.dp_flush()
        flags = int(self._steps == 0) | (2 if nesterov else 0)
        _lib.check(self.lib.pevit_sgd_step(self._ctx, _lib.stream_ptr(), lr, momentum, weight_decay, grad_scale, flags),
                   "pevit_sgd_step")
        # the fused SGD kernel skips the update on device when a stream-K hand-off of this step timed out; the host reads
        # the error word on the first step and then every STREAMK_CHECK_EVERY steps (each check synchronises the stream)
        # ... and with the flat all-reduce as the exchange EVERY step: a rank whose reduction gave up withholds its update while its
        # peers apply theirs, and replicas must not drift apart for 64 steps before anybody notices
        if self._steps % self.STREAMK_CHECK_EVERY == 0 or getattr(self, "_flat_ar", None) is not None:
            self.check_streamk()
        self._steps += 1

    STREAMK_CHECK_EVERY = 64

    def check_streamk(self):
        """Stream-K GEMMs hand partial tiles between workgroups inside one launch; a consumer that never saw its
        producer gives up after ~1 s and raises an error word instead of hanging.  Fails loudly if that happened, and says how
        many optimizer updates the fused SGD kernel withheld since.  The word is NOT cleared here: every later update is
        withheld and every later check raises again until the caller acknowledges with ``clear_streamk_error()``."""
        word, skipped = C.c_uint(), C.c_uint()
        _lib.check(self.lib.pevit_streamk_status(self._ctx, _lib.stream_ptr(), C.byref(word), C.byref(skipped)), "pevit_streamk_status")
        if getattr(self, "_flat_ar", None) is not None:       # the DP exchange's error word (updates were withheld on device while raised)
            self._flat_ar.check(stream=self._flat_ar_stream)
        if word.value:
            raise _lib.PevitError(f"stream-K GEMM hand-off timed out: the logits / loss returned since are invalid and "
                                  f"{skipped.value} optimizer update(s) were withheld on device (parameters and momentum are those "
                                  f"of the last good step); call clear_streamk_error() to resume")

    def clear_streamk_error(self) -> bool:
        """Acknowledge a stream-K error (clears the device word and the withheld-update counter); True if one was pending."""
        rc = self.lib.pevit_streamk_error(self._ctx, _lib.stream_ptr())
        if rc < 0:
            raise _lib.PevitError("pevit_streamk_error failed")
        return rc > 0

    PROF_KINDS = {0: "ln_fwd", 1: "ln_bwd", 2: "attn_fwd", 3: "attn_bwd", 4: "delta_add", 5: "lowrank_u", 6: "lowrank_grad",
                  7: "im2col", 8: "lowrank_bwd", 9: "attn_fwd_delta", 10: "adapter_fwd", 11: "adapter_bwd"}     # include/pevit_hip.h: enum pevit_prof_kind

    def profile_gemms(self, fn, max_launches=8192, all_kernels=False):
        """Run ``fn()`` with HIP events around every GEMM launch; returns (ms, flops, launches) of the GEMM family; the
        algorithmic operand+result bytes of those launches are left in ``self.last_profile_bytes``, the per-shape table in
        ``self.last_profile_by_shape``.  ``all_kernels``: the HBM-bound kernels of the step are bracketed too and land in
        ``self.last_profile_hbm`` = {kernel: [launches, ms, algorithmic bytes]}."""
        self.tune("profile_all", int(all_kernels))
        try:
            _lib.check(self.lib.pevit_profile_begin(self._ctx, max_launches), "pevit_profile_begin")
            fn()
            ms, fl, by, n = C.c_double(), C.c_double(), C.c_double(), C.c_int()
            _lib.check(self.lib.pevit_profile_end(self._ctx, C.byref(ms), C.byref(fl), C.byref(by), C.byref(n)),
                       "pevit_profile_end")
        finally:
            self.tune("profile_all", 0)
        self.last_profile_bytes = by.value
        # per launch: (epilogue, M, N, K) -> [launches, ms, flops, bytes]
        per, hbm, n_gemm = {}, {}, 0
        one_ms, one_fl, one_by, shape = C.c_double(), C.c_double(), C.c_double(), (C.c_int * 4)()
        for i in range(n.value):
            _lib.check(self.lib.pevit_profile_launch(self._ctx, i, C.byref(one_ms), C.byref(one_fl), shape), "pevit_profile_launch")
            _lib.check(self.lib.pevit_profile_launch_bytes(self._ctx, i, C.byref(one_by)), "pevit_profile_launch_bytes")
            if shape[0] >= 100:
                e = hbm.setdefault(self.PROF_KINDS.get(shape[0] - 100, str(shape[0] - 100)), [0, 0.0, 0.0])
                e[0] += 1; e[1] += one_ms.value; e[2] += one_by.value
            else:
                n_gemm += 1
                e = per.setdefault(tuple(shape), [0, 0.0, 0.0, 0.0])
                e[0] += 1; e[1] += one_ms.value; e[2] += one_fl.value; e[3] += one_by.value
        self.last_profile_by_shape = per
        self.last_profile_hbm = hbm
        return ms.value, fl.value, n_gemm

    def reset_optimizer(self):
        self.dp_flush()
        self._steps = 0
        self.momentum.zero_()

    def reset_run(self):
        """State a fresh ``Classifier`` would have, with the frozen backbone left resident (the
        reference rebuilds everything for each of its ~90 sweep runs)."""
        self.dp_flush()
        self.reset_optimizer()
        self.params.zero_(); self.grads.zero_()
        self.running_mean.zero_(); self.running_var.fill_(1.0)

    def train_step(self, images, labels, lr, momentum=0.9, weight_decay=0.0, bn_training=True, process_group=None,
                   world_size=1, nesterov=False, logits_out=None, loss_out=None):
        """One reference ``train_one`` iteration (kadaptation_clip.py:347-353).  With DP the flat gradient buffer
        is the only thing that crosses xGMI (the frozen backbone never does); how it is exchanged is ``dp_exchange_mode``
        (default: one in-stream all-reduce behind the fused forward/backward call, 1/world folded into the SGD kernel)."""
        if world_size <= 1:
            logits, loss = self.forward_backward(images, labels, bn_training, logits_out, loss_out)
            self.sgd_step(lr, momentum, weight_decay, 1.0, nesterov)
            return logits, loss
        if self.dp_exchange_mode == "pipelined":
            return self._train_step_pipelined(images, labels, lr, momentum, weight_decay, bn_training, process_group, world_size,
                                              nesterov, logits_out, loss_out)
        logits, loss = self.forward_backward_dp(images, labels, bn_training, process_group)
        if logits_out is not None:
            logits_out.copy_(logits)
        if loss_out is not None:
            loss_out.copy_(loss)
        self.sgd_step(lr, momentum, weight_decay, 1.0 / world_size, nesterov)
        return logits, loss

    # ---- DP with the exchange under the next step's stem (round 5) ----------------------------------------------------------------
    def _train_step_pipelined(self, images, labels, lr, momentum, weight_decay, bn_training, process_group, world_size, nesterov,
                              logits_out, loss_out):
        """``dp_exchange_mode = "pipelined"``: the fused forward/backward call on the current stream; the all-reduce of the flat
        gradient buffer and the fused SGD kernel on a SECOND stream behind it; the next fused call runs its stem (patch gather,
        patch embedding, ln_pre: ~70 us that read no trainable parameter) and only then waits, on the device, for the update
        (``pevit_set_step_gate``).  The exchange's latency and the cross-stream hand-overs around the collective leave the critical
        path.  Same kernels, same order of arithmetic as the other modes: bit-identical results (tests/test_gpu_dp.py).
        The caller's obligations: every step of a run goes through ``train_step`` on ONE stream; ``dp_flush()`` before anything
        else reads parameters / momentum / gradients or launches on them (validation, ``state_dict``, another mode) -- the
        engine's own entry points that do so call it."""
        cur = torch.cuda.current_stream(self.device)
        st = getattr(self, "_pipe", None)
        if st is None:
            st = self._pipe = {"stream": torch.cuda.Stream(self.device), "grads": torch.cuda.Event(), "params": torch.cuda.Event(),
                               "open": False}
            st["params"].record(cur)                       # (creates the event: nothing to wait for before the first step)
            _lib.check(self.lib.pevit_set_step_gate(self._ctx, C.c_void_p(st["params"].cuda_event)), "pevit_set_step_gate")
        st["open"] = True
        logits, loss = self.forward_backward(images, labels, bn_training, logits_out, loss_out)     # waits for st["params"] behind its stem
        st["grads"].record(cur)
        with torch.cuda.stream(st["stream"]):
            st["stream"].wait_event(st["grads"])
            xt = self._xt_pair()
            if xt: xt[0].record(st["stream"])
            self._exchange(self.grads, process_group, in_stream=True).wait()
            if xt: xt[1].record(st["stream"])
            self.sgd_step(lr, momentum, weight_decay, 1.0 / world_size, nesterov, _pipelined=True)
            st["params"].record(st["stream"])
        return logits, loss

    # ---- how long the exchange takes on the device (bench.py N > 1: `exchange_us_per_step`) ---------------------------------------
    def time_exchange(self, on: bool = True):
        """Bracket every gradient exchange of the DP step with a pair of events on the stream it is enqueued on (single / staged:
        the compute stream -- for staged the pair brackets the WAITS at the end of the backward, i.e. the part of the three
        asynchronous all-reduces that the backward did not hide; pipelined: the second stream).  ``exchange_times_us()`` reads
        them back."""
        self._xt = [] if on else None

    def _xt_pair(self):
        if getattr(self, "_xt", None) is None:
            return None
        pair = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        self._xt.append(pair)
        return pair

    def exchange_times_us(self):
        """Microseconds of every bracketed exchange since time_exchange(True) (synchronises the device)."""
        torch.cuda.synchronize(self.device)
        return [a.elapsed_time(b) * 1e3 for a, b in (getattr(self, "_xt", None) or [])]

    def dp_flush(self):
        """Pipelined DP: make the current stream wait for the last exchange + update (no-op otherwise)."""
        st = getattr(self, "_pipe", None)
        if st is not None and st["open"]:
            torch.cuda.current_stream(self.device).wait_event(st["params"])
            st["open"] = False

    def dp_pipeline_off(self):
        """Leave the pipelined schedule: flush, detach the step gate from the context, drop the second stream."""
        self.dp_flush()
        if getattr(self, "_pipe", None) is not None:
            _lib.check(self.lib.pevit_set_step_gate(self._ctx, None), "pevit_set_step_gate")
            self._pipe = None

    # ---- whole-step HIP graph -----------------------------------------------------
    def capture_train_step(self, images, labels, lr, momentum=0.9, weight_decay=0.0, bn_training=True, nesterov=False):
        """Capture ``forward_backward + sgd_step`` on THESE buffers (pointers, batch, hyper-parameters and BatchNorm mode are baked
        in) into a HIP graph and return a callable that replays it: one hipGraphLaunch instead of ~200 kernel launches from the
        C call.  The step counter, the periodic error-word check and the results (``self._logits``, ``self._loss``) behave as in
        ``train_step``.  At least one eager step must have run (the first SGD step has no momentum buffer to read, and the
        library's one-time attribute calls are not stream operations).  Same kernels, same order: bit-identical to eager
        (tests/test_gpu_mirror.py::test_graph_replay_equals_eager).  Measured: profiles/r05_graph_capture.md."""
        if self._steps == 0:
            raise _lib.PevitError("capture_train_step: run one eager train_step first (first-step flag, one-time kernel attributes)")
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            raise _lib.PevitError("capture_train_step is a single-GPU tool: the captured step has no gradient exchange")
        # leaves the pipelined schedule altogether: a step gate still attached to the context would bake a wait on an event
        # recorded OUTSIDE the capture (and the gated zero_grads path) into the graph
        self.dp_pipeline_off()
        B = images.shape[0]
        self._check_batch(images, labels)
        fn = self.lib.pevit_train_forward_backward_u8 if images.dtype == torch.uint8 else self.lib.pevit_train_forward_backward
        flags = 2 if nesterov else 0
        logits, loss = self._logits[:B], self._loss
        graph = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream(self.device)
        side.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(side):
            with torch.cuda.graph(graph, stream=side, capture_error_mode="thread_local"):
                _lib.check(fn(self._ctx, _lib.stream_ptr(), _lib.ptr(images), _lib.ptr(labels), _lib.ptr(self.running_mean),
                              _lib.ptr(self.running_var), int(bn_training), _lib.ptr(logits), _lib.ptr(loss), B),
                           "pevit_train_forward_backward (capture)")
                _lib.check(self.lib.pevit_sgd_step(self._ctx, _lib.stream_ptr(), lr, momentum, weight_decay, 1.0, flags),
                           "pevit_sgd_step (capture)")
        torch.cuda.current_stream(self.device).wait_stream(side)
        keep = (images, labels)                     # the graph reads these addresses on every replay

        def replay():
            self.forward_generation += 1
            graph.replay()
            if self._steps % self.STREAMK_CHECK_EVERY == 0:
                self.check_streamk()
            self._steps += 1
            return logits, loss
        replay.graph, replay.buffers = graph, keep
        return replay

    def use_flat_allreduce(self, process_group=None):
        """Exchange the gradient buckets with pevit_allreduce_flat (dp.FlatAllReduce: IPC-mapped peer mailboxes, copy-engine
        pushes, deterministic local reduction on a side stream) instead of the process group's all-reduce."""
        from . import dp
        self._flat_ar = dp.FlatAllReduce(self.n_params, group=process_group, device=self.device)
        self._flat_ar_stream = torch.cuda.Stream(self.device)
        # a bucket whose exchange gave up (a peer never arrived, sizes disagree) must not reach the parameters: the fused SGD
        # kernel reads the all-reduce's error word next to the stream-K one and withholds the update while either is raised
        _lib.check(self.lib.pevit_set_external_poison(self._ctx, C.c_void_p(self._flat_ar.error_word())), "pevit_set_external_poison")
        return self._flat_ar

    def _exchange(self, view, process_group, in_stream=False):
        """Start the sum all-reduce of one gradient bucket; returns something with .wait().
        in_stream: nothing is to run beside the exchange (the single-exchange schedule): the SYNCHRONOUS form of the collective,
        which this torch enqueues on the CURRENT stream.  Measured on a 1-rank RCCL group (scripts/r5_dp_hosttime.py): a round trip
        through a second stream -- what ``async_op=True`` + ``wait()`` is, with or without a collective on it -- costs 74 us of
        device time per step on this platform, the synchronous call 7 us."""
        import torch.distributed as dist
        ar = getattr(self, "_flat_ar", None)
        if ar is None and in_stream:
            dist.all_reduce(view, op=dist.ReduceOp.SUM, group=process_group)

            class _Done:
                def wait(_self):
                    pass
            return _Done()
        if ar is None:
            return dist.all_reduce(view, op=dist.ReduceOp.SUM, group=process_group, async_op=True)
        if in_stream:
            ar.all_reduce(view, stream=torch.cuda.current_stream(self.device))

            class _Done2:
                def wait(_self):
                    pass
            return _Done2()
        cur, side = torch.cuda.current_stream(self.device), self._flat_ar_stream
        ready = torch.cuda.Event(); ready.record(cur)
        side.wait_event(ready)                          # the bucket's gradients are final
        ar.all_reduce(view, stream=side)
        done = torch.cuda.Event(); done.record(side)

        class _W:
            def wait(_self):
                torch.cuda.current_stream(self.device).wait_event(done)
        return _W()

    #: how the DP step exchanges the gradients: "single" = the fused forward/backward call, then ONE all-reduce of the whole
    #: flat buffer (406 KB for KAdaptation) | "staged" = the backward cut into stages with three overlapped buckets (rounds 2-4) |
    #: "pipelined" (train_step only) = "single" with the exchange + SGD on a second stream under the NEXT step's stem.
    #: Round 5 made "single" the default: the exchange is latency-bound at this size, so what the staged route can hide is the
    #: transfer of two of its three buckets, while the LAST bucket's latency is exposed either way -- and the staging itself (a
    #: second reduce / chain / rule-sum group, three round trips through the collective's stream at ~74 us each, stream-K off)
    #: costs 3.6 % of the step on one rank against 0.2 % for one in-stream exchange behind the fused call
    #: (bench.py --dp-route; profiles/r05_dp_single_exchange.md).
    dp_exchange_mode = "single"

    def forward_backward_dp(self, images, labels, bn_training=True, process_group=None, mode=None):
        """forward_backward() followed by the gradient exchange; leaves the SUM over ranks in ``self.grads``.
        mode "single" (default, ``dp_exchange_mode``): the fused call, then one all-reduce of the flat gradient buffer.
        mode "staged": the call cut into stages so that the exchange overlaps the backward -- three buckets of the flat buffer, each
        all-reduced asynchronously as soon as its gradients are final:  head (after the head backward) | blocks L/2..L-1 (after
        the upper half of the tower backward) | shared rules + blocks 0..L/2-1 (at the end).  Same kernels in the same order as
        the fused call either way, so a single rank reproduces forward_backward() bit for bit."""
        mode = mode or self.dp_exchange_mode
        self.dp_flush()
        if mode == "pipelined":
            raise ValueError("forward_backward_dp: the pipelined schedule spans the optimizer step -- use train_step")
        if mode == "single":
            if getattr(self, "_dp_streamk_off", False):        # (a staged step before this one switched it off)
                self.tune("gemm_streamk", 1); self._dp_streamk_off = False
            logits, loss = self.forward_backward(images, labels, bn_training)
            xt = self._xt_pair()
            if xt: xt[0].record()
            self._exchange(self.grads, process_group, in_stream=True).wait()
            if xt: xt[1].record()
            return logits, loss
        if mode != "staged":
            raise ValueError(f"forward_backward_dp: unknown mode {mode!r}")
        import torch.distributed as dist
        B = images.shape[0]
        self._check_batch(images, labels)
        if not getattr(self, "_dp_streamk_off", False):
            # the RCCL kernels of the overlapped all-reduce share the CUs with the second half of the backward: stream-K's
            # hand-off assumes all its workgroups are co-resident, so under DP the plain tilings are used instead
            self.tune("gemm_streamk", 0)
            self._dp_streamk_off = True
        self.zero_grad()
        feat = self.visual_forward(images, save=True)
        _lib.check(self.lib.pevit_head_forward_backward(
            self._ctx, _lib.stream_ptr(), _lib.ptr(feat), _lib.ptr(labels), _lib.ptr(self.running_mean),
            _lib.ptr(self.running_var), int(bn_training), _lib.ptr(self._logits), _lib.ptr(self._loss),
            _lib.ptr(self._dfeat(B)), B), "pevit_head_forward_backward")
        L = self.arch.layers
        mid = L // 2
        cut = self.lib.pevit_param_layer_offset(self._ctx, mid)
        work = [self._exchange(self.grads[self.n_tower:], process_group)]
        if self.n_tower == 0:                                  # frozen tower (linear probe): nothing below the head trains
            work[0].wait()
            return self._logits[:B], self._loss
        dfeat = self._dfeat(B)
        if mid > 0:
            _lib.check(self.lib.pevit_visual_backward_part(self._ctx, _lib.stream_ptr(), _lib.ptr(dfeat), B, L, mid),
                       "pevit_visual_backward_part")
            work.append(self._exchange(self.grads[cut:self.n_tower], process_group))
            _lib.check(self.lib.pevit_visual_backward_part(self._ctx, _lib.stream_ptr(), None, B, mid, 0),
                       "pevit_visual_backward_part")
            work.append(self._exchange(self.grads[:cut], process_group))
        else:
            self.visual_backward(dfeat)
            work.append(self._exchange(self.grads[:self.n_tower], process_group))
        xt = self._xt_pair()
        if xt: xt[0].record()
        for w in work:
            w.wait()
        if xt: xt[1].record()
        return self._logits[:B], self._loss

    def sync_replicas(self, process_group=None, src: int = 0):
        """Make every rank's trainable state identical to rank ``src``'s (parameters, momentum, BatchNorm running
        statistics, step counter semantics): call once after the engines have been loaded and before the first DP step."""
        self.dp_flush()
        import torch.distributed as dist
        for t in (self.params, self.momentum, self.running_mean, self.running_var):
            dist.broadcast(t, src=src, group=process_group)

    def average_bn_buffers(self, process_group=None):
        """BatchNorm running statistics follow the LOCAL shards; average them across ranks before validation or a
        checkpoint (dp.average_bn_buffers)."""
        self.dp_flush()
        dp.average_bn_buffers(self.running_mean, self.running_var, process_group)

    def _dfeat(self, B):
        if getattr(self, "_dfeat_buf", None) is None or self._dfeat_buf.shape[0] < B:
            self._dfeat_buf = torch.empty((max(B, self.max_batch), self.arch.embed_dim), dtype=torch.float32, device=self.device)
        return self._dfeat_buf[:B]
