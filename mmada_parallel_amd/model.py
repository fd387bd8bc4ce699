"""Host-side mirror of the reference model contract, backed by libmmada_mi355x.so.

`LLaDAForMultiModalGeneration` keeps the surface `generate_ti2ti` / `inference.py` use:
    model = LLaDAForMultiModalGeneration.from_pretrained(path, torch_dtype=torch.bfloat16, device_map="auto")
    model(ids, infer=True, use_cache=False).logits        # [B, L, V] bf16
    model.config.text_vocab_size / codebook_size ; model.device
(reference: model/modeling_xllmx_dimoo.py:24-72, model/modeling_llada.py:1462-1511, inference.py:83-89).

Extra fast-path methods (`forward_body`, `head_rows`) let the accelerated sampler avoid materialising [L, V] logits.
PyTorch is used for device memory, streams and (TP) torch.distributed only.
"""
from __future__ import annotations

import ctypes as C
import json
import os
from types import SimpleNamespace
from typing import Dict, Optional

import torch

from . import abi

_SUPPORTED = dict(block_type="llama", activation_type="silu", layer_norm_type="rms")
# what the reference's ModelConfig assumes for a key that is absent from config.json (model/configuration_llada.py:147-317):
# a trimmed config must be read the way the reference would read it, not silently as the llama/silu/untied layout
_REFERENCE_DEFAULTS = dict(block_type="sequential", activation_type="swiglu", layer_norm_type="default", rope=False,
                           rope_full_precision=True, weight_tying=True, include_bias=False, alibi=False,
                           attention_layer_norm=False, scale_logits=False, input_emb_norm=False, include_qkv_bias=None,
                           rms_norm_eps=1e-5, rope_theta=10000.0, max_sequence_length=1024, mlp_ratio=4,
                           multi_query_attention=None, n_kv_heads=None)


class CausalLMOutputLite(SimpleNamespace):
    """Minimal stand-in for transformers' CausalLMOutputWithPast: only `.logits` is consumed on this path."""


class LLaDAConfigLite(SimpleNamespace):
    def get(self, k, default=None):
        return getattr(self, k, default)

    def ref(self, k):
        """Value of a ModelConfig field, falling back to the REFERENCE's default when the key is absent."""
        return getattr(self, k, _REFERENCE_DEFAULTS[k])


def effective_n_kv_heads(cfg: LLaDAConfigLite) -> int:
    """ModelConfig.effective_n_kv_heads (model/configuration_llada.py:366-384)."""
    n_kv, mqa = cfg.ref("n_kv_heads"), cfg.ref("multi_query_attention")
    if n_kv is None:
        return 1 if mqa is True else cfg.n_heads
    if mqa is None:
        return n_kv
    should = 1 if mqa else cfg.n_heads
    if n_kv != should:
        raise ValueError("You can't set `multi_query_attention` and `n_kv_heads` at the same time.")
    return should


def _validate(cfg: LLaDAConfigLite) -> None:
    def _name(v):
        return getattr(v, "value", v)

    for k, want in _SUPPORTED.items():
        got = _name(cfg.ref(k))
        if str(got) != want:
            how = "" if hasattr(cfg, k) else " (the reference's default for a key missing from the config)"
            raise NotImplementedError(f"config.{k}={got!r}{how}: only {want!r} is on the MI355X hot path")
    for flag in ("alibi", "attention_layer_norm", "scale_logits", "input_emb_norm", "include_bias", "include_qkv_bias"):
        if cfg.ref(flag):
            raise NotImplementedError(f"config.{flag}=True is not supported on the MI355X hot path")
    if not cfg.ref("rope") or not cfg.ref("rope_full_precision"):
        raise NotImplementedError("rope=True and rope_full_precision=True are required (reference default when the key "
                                  "is missing: rope=False)")
    if cfg.d_model // cfg.n_heads != 128:
        raise NotImplementedError("head_dim must be 128")


class LLaDAForMultiModalGeneration:
    """MI355X-native drop-in for the reference class of the same name (inference path only)."""

    MASK_TOKEN = 126336

    def __init__(self, config, state_dict: Dict[str, torch.Tensor], device: Optional[torch.device] = None,
                 tp_rank: int = 0, tp_size: int = 1, max_batch: int = 3, max_seq: Optional[int] = None):
        if isinstance(config, dict):
            config = LLaDAConfigLite(**config)
        _validate(config)
        if not torch.cuda.is_available():
            raise abi.MmadaError("LLaDAForMultiModalGeneration needs an MI355X (torch.cuda unavailable); "
                                 "there is no CPU fallback")
        self.config = config
        self.device = torch.device(device if device is not None else f"cuda:{torch.cuda.current_device()}")
        self.dtype = torch.bfloat16
        self.tp_rank, self.tp_size = tp_rank, tp_size
        self._lib = abi.lib()
        self._handle = C.c_void_p()
        self._ws = None
        self._ws1 = None
        self._ws_bytes = [0, 0]  # bytes registered with the library per activation context
        self._ws_epoch = 0       # bumped whenever a workspace is (re)allocated: captured step graphs hold its addresses
        self.graph_replays, self.graph_nodes = 0, {}  # hipGraph step replays issued / nodes per captured step kind
        self._comm_in_library, self._comm_rows, self.tp_collective = False, 0, None
        self._handle1 = None
        self._split = None
        self.n_kv_heads = effective_n_kv_heads(config)
        self.vocab = config.get("embedding_size") or config.vocab_size
        self.mlp_hidden = config.get("mlp_hidden_size") or config.ref("mlp_ratio") * config.d_model
        self.max_seq = max_seq or max(int(config.ref("max_sequence_length")), 4096)
        self.max_batch = max_batch

        c = abi.MmadaCfg(
            d_model=config.d_model, n_layers=config.n_layers, n_heads=config.n_heads, n_kv_heads=self.n_kv_heads,
            head_dim=128, mlp_hidden=self.mlp_hidden, vocab=self.vocab, max_seq=self.max_seq,
            rms_eps=float(config.ref("rms_norm_eps")), rope_theta=float(config.ref("rope_theta")),
            tp_rank=tp_rank, tp_size=tp_size, mask_token_id=int(config.get("mask_token_id", self.MASK_TOKEN)),
            text_vocab_size=int(config.get("text_vocab_size", 126356)),
            codebook_size=int(config.get("codebook_size", 8192)), reserved=0)
        # inv_freq exactly as RotaryEmbedding.get_rotary_embedding computes it (model/modeling_llada.py:391-393)
        inv_freq = 1.0 / (c.rope_theta ** (torch.arange(0, 128, 2, dtype=torch.float) / 128))
        inv = (C.c_float * 64)(*inv_freq.tolist())
        with torch.cuda.device(self.device):
            abi.check(self._lib.mmada_create(C.byref(c), inv, C.byref(self._handle)), "mmada_create")
            self._bind(state_dict)

    # ---- loading ------------------------------------------------------------------------------------------------
    def _bind(self, sd: Dict[str, torch.Tensor]) -> None:
        p = "model.transformer."
        dev, dt = self.device, torch.bfloat16

        def get(name):
            if name not in sd:
                raise KeyError(f"checkpoint is missing {name}")
            return sd[name].to(device=dev, dtype=dt).contiguous()

        self._wte = get(p + "wte.weight")
        self._ln_f = get(p + "ln_f.weight")
        self._head = self._wte if self.config.ref("weight_tying") else get(p + "ff_out.weight")
        abi.check(self._lib.mmada_bind_globals(self._handle, self._wte.data_ptr(), self._ln_f.data_ptr(),
                                               self._head.data_ptr()), "mmada_bind_globals")
        st = abi.stream_ptr()
        for i in range(self.config.n_layers):
            b = f"{p}blocks.{i}."
            names = ["attn_norm", "ff_norm", "q_proj", "k_proj", "v_proj", "attn_out", "ff_proj", "up_proj", "ff_out"]
            ts = [get(b + n + ".weight") for n in names]
            abi.check(self._lib.mmada_bind_layer(self._handle, i, *[t.data_ptr() for t in ts], st), "mmada_bind_layer")
            torch.cuda.current_stream().synchronize()  # originals may be freed once the repack has drained
            del ts

    @classmethod
    def from_state_dict(cls, config, state_dict, **kw):
        return cls(config, state_dict, **kw)

    @classmethod
    def from_pretrained(cls, path: str, torch_dtype=torch.bfloat16, device_map="auto", **kw):
        """Loads config.json + *.safetensors with the reference's state-dict keys (checkpoints drop in unchanged)."""
        from safetensors import safe_open

        if torch_dtype not in (None, torch.bfloat16):
            raise NotImplementedError("the MI355X hot path computes in bf16")
        with open(os.path.join(path, "config.json")) as f:
            config = LLaDAConfigLite(**json.load(f))
        files = sorted(f for f in os.listdir(path) if f.endswith(".safetensors"))
        if not files:
            raise FileNotFoundError(f"no *.safetensors under {path}")

        class _Lazy(dict):
            def __init__(self):
                super().__init__()
                self._where = {}
                for fn in files:
                    with safe_open(os.path.join(path, fn), "pt") as sf:
                        for k in sf.keys():
                            self._where[k] = fn

            def __contains__(self, k):
                return k in self._where

            def __getitem__(self, k):
                with safe_open(os.path.join(path, self._where[k]), "pt") as sf:
                    return sf.get_tensor(k)

        return cls(config, _Lazy(), **kw)

    # ---- workspace ------------------------------------------------------------------------------------------------
    def _ensure_ws(self, B: int, L: int, lane: int = 0) -> None:
        h = self._lane_handle(lane)
        need = self._lib.mmada_workspace_bytes(h, B, L)
        if need > self._ws_bytes[lane]:  # compare with what the LIBRARY was given, not with the padded tensor
            grow = max(need, self._lib.mmada_workspace_bytes(h, max(B, self.max_batch), L))
            ws = torch.empty(grow + 256, dtype=torch.uint8, device=self.device)
            base = (ws.data_ptr() + 255) // 256 * 256
            abi.check(self._lib.mmada_set_workspace(h, base, grow), "mmada_set_workspace")
            self._ws_bytes[lane] = grow
            self._ws_epoch += 1
            if lane == 0:
                self._ws = ws
            else:
                self._ws1 = ws

    def _lane_handle(self, lane: int):
        if lane == 0:
            return self._handle
        if getattr(self, "_handle1", None) is None:
            self._handle1 = C.c_void_p()
            self._ws1 = None
            abi.check(self._lib.mmada_clone_shared(self._handle, C.byref(self._handle1)), "mmada_clone_shared")
        return self._handle1

    # ---- forward ---------------------------------------------------------------------------------------------------
    def forward_body(self, input_ids: torch.Tensor, consumed: Optional[tuple] = None) -> None:
        """Embedding + all blocks; the final residual stream stays resident for head_rows().

        consumed = (row_begin, row_end): the caller promises to read only rows [row_begin, row_end) of each sequence
        through head_rows(); the last block then skips the other rows (mmada_set_consumed_rows; bit-identical on the
        consumed rows).  None = every row.

        Tensor parallel (tp_size > 1): after each of the two row-parallel GEMMs of a block the partial residual
        stream is all-reduced over RCCL.  With B >= 2 the batch is split into two micro-batches living in two
        activation contexts over the same weights, and the all-reduce of one micro-batch is issued asynchronously
        so that it overlaps the other micro-batch's GEMMs / attention."""
        ids = input_ids.to(device=self.device, dtype=torch.long).contiguous()
        B, L = ids.shape
        st = abi.stream_ptr()
        in_lib = self.tp_size > 1 and self._comm_in_library
        microbatch = B >= 2 and not in_lib and (self.tp_size > 1 or os.environ.get("MMADA_MICROBATCH") == "1")
        self._split = None
        lo, hi = (int(consumed[0]), int(consumed[1])) if consumed is not None else (0, 0)
        if consumed is not None and not 0 <= lo < hi <= L:
            raise ValueError(f"consumed rows {consumed} outside [0, {L})")
        self._consumed = (lo, hi) if consumed is not None else None
        for lane in ((0, 1) if microbatch else (0,)):
            abi.check(self._lib.mmada_set_consumed_rows(self._lane_handle(lane), lo, hi), "mmada_set_consumed_rows")
        if not microbatch:
            self._ensure_ws(B, L)
            if self.tp_size == 1 or in_lib:
                # tp_size > 1: the reduce-scatter / RMSNorm / all-gather exchanges are issued by the library (tp_comm.hip)
                if in_lib and B * ((L + 7) // 8 * 8) > self._comm_rows:
                    raise abi.MmadaError(f"forward of {B}x{L} exceeds the {self._comm_rows} rows init_tp_comm() was sized for")
                abi.check(self._lib.mmada_forward_body(self._handle, ids.data_ptr(), B, L, st), "mmada_forward_body")
            else:
                import torch.distributed as dist

                abi.check(self._lib.mmada_embed(self._handle, ids.data_ptr(), B, L, st), "mmada_embed")
                for i in range(self.config.n_layers):
                    for seg in (self._lib.mmada_attn_partial, self._lib.mmada_mlp_partial):
                        abi.check(seg(self._handle, i, st), "mmada_*_partial")
                        dist.all_reduce(self._stream_view())
        else:
            import torch.distributed as dist

            reduce = dist.is_available() and dist.is_initialized()
            B0 = (B + 1) // 2
            parts = [ids[:B0].contiguous(), ids[B0:].contiguous()]
            handles = [self._lane_handle(0), self._lane_handle(1)]
            for j in (0, 1):
                self._ensure_ws(parts[j].shape[0], L, lane=j)
                abi.check(self._lib.mmada_embed(handles[j], parts[j].data_ptr(), parts[j].shape[0], L, st), "mmada_embed")
            pending = [None, None]
            for i in range(self.config.n_layers):
                for seg in (self._lib.mmada_attn_partial, self._lib.mmada_mlp_partial):
                    for j in (0, 1):
                        if pending[j] is not None:
                            pending[j].wait()  # this lane's previous all-reduce (ran under the other lane's kernels)
                            pending[j] = None
                        abi.check(seg(handles[j], i, st), "mmada_*_partial")
                        if reduce:
                            pending[j] = dist.all_reduce(self._stream_view(j), async_op=True)
            for j in (0, 1):
                if pending[j] is not None:
                    pending[j].wait()
            self._split = B0
        self._shape = (B, L)

    def _stream_view(self, lane: int = 0) -> torch.Tensor:
        """Torch view of the library's current residual-stream buffer (inside our workspace tensor)."""
        h = self._lane_handle(lane)
        ws = self._ws if lane == 0 else self._ws1
        p = self._lib.mmada_stream_ptr(h)
        n = self._lib.mmada_stream_bytes(h)
        off = p - ws.data_ptr()
        return ws[off:off + n].view(torch.bfloat16)

    def head_rows(self, rows: torch.Tensor, col_begin: int, col_end: int, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """logits[r] = lm_head[col_begin:col_end] · ln_f(x[rows[r]]), rows = b*L + l (int32, device).

        `rows` must be batch-major with the same number of rows per batch element (what generate_ti2ti builds).
        `out` (bf16 [R, col_end - col_begin], contiguous): write there instead of allocating (fixed address: capturable)."""
        rows = rows.to(device=self.device, dtype=torch.int32).contiguous()
        if os.environ.get("MMADA_CHECK_ROWS") == "1" and getattr(self, "_consumed", None) is not None and rows.numel():
            l = rows % self._shape[1]  # debug aid (forces a device sync): rows must lie inside the declared window
            assert int(l.min()) >= self._consumed[0] and int(l.max()) < self._consumed[1], "head_rows outside forward_body(consumed=...)"
        if out is None:
            out = torch.empty((rows.numel(), col_end - col_begin), dtype=torch.bfloat16, device=self.device)
        elif out.shape != (rows.numel(), col_end - col_begin) or out.dtype != torch.bfloat16 or not out.is_contiguous():
            raise ValueError("head_rows(out=...): need a contiguous bf16 [rows, col_end - col_begin] tensor")
        st = abi.stream_ptr()
        if getattr(self, "_split", None) is None:
            abi.check(self._lib.mmada_head_rows(self._handle, rows.data_ptr(), rows.numel(), col_begin, col_end,
                                                out.data_ptr(), st), "mmada_head_rows")
            return out
        B, L = self._shape
        if rows.numel() % B:
            raise ValueError("head_rows on a micro-batched forward needs an equal row count per batch element")
        per_b, B0 = rows.numel() // B, self._split
        cut = B0 * per_b
        r1 = (rows[cut:] - B0 * L).contiguous()  # second micro-batch indexes its own batch from 0
        abi.check(self._lib.mmada_head_rows(self._handle, rows.data_ptr(), cut, col_begin, col_end, out.data_ptr(), st),
                  "mmada_head_rows")
        abi.check(self._lib.mmada_head_rows(self._handle1, r1.data_ptr(), rows.numel() - cut, col_begin, col_end,
                                            out[cut:].data_ptr(), st), "mmada_head_rows")
        return out

    def hidden_state(self) -> torch.Tensor:
        """Residual stream after the last block, [B, L, d] (parity tap)."""
        B, L = self._shape
        out = torch.empty((B, L, self.config.d_model), dtype=torch.bfloat16, device=self.device)
        st = abi.stream_ptr()
        if getattr(self, "_split", None) is None:
            abi.check(self._lib.mmada_read_stream(self._handle, out.data_ptr(), st), "mmada_read_stream")
        else:
            abi.check(self._lib.mmada_read_stream(self._handle, out.data_ptr(), st), "mmada_read_stream")
            abi.check(self._lib.mmada_read_stream(self._handle1, out[self._split:].data_ptr(), st), "mmada_read_stream")
        return out

    def debug_buffer(self, which: int) -> torch.Tensor:
        """Parity tap (tests only): flat bf16 view of an intermediate of the most recent block, see mmada_debug_buffer."""
        p, lp, lkv = C.c_void_p(), C.c_int32(), C.c_int32()
        abi.check(self._lib.mmada_debug_buffer(self._handle, which, C.byref(p), C.byref(lp), C.byref(lkv)), "debug_buffer")
        B, L = self._shape
        d, F = self.config.d_model, self.mlp_hidden // self.tp_size
        hq, hkv = self.config.n_heads // self.tp_size, self.n_kv_heads // self.tp_size
        shapes = {0: (B * lp.value, d), 1: (B, hq, lkv.value, 128), 2: (B, hkv, lkv.value, 128),
                  3: (B, hkv, 128, lkv.value), 4: (B * lp.value, hq * 128), 5: (B * lp.value, F)}
        shape = shapes[which]
        n = 2
        for v in shape:
            n *= v
        off = p.value - self._ws.data_ptr()
        t = self._ws[off:off + n].view(torch.bfloat16).view(*shape)
        if which == 3:  # undo the [0,4,1,5,2,6,3,7] chunk order of every 32-key block (csrc/common.h vt_key_pos)
            t = t.reshape(B, hkv, 128, lkv.value // 32, 8, 4)[..., [0, 2, 4, 6, 1, 3, 5, 7], :].reshape(B, hkv, 128, lkv.value)
        return t

    def forward(self, input_ids=None, labels=None, infer=False, use_cache=False, to_compute_mask=None, cat="", **_):
        """LLaDAForMultiModalGeneration.forward(infer=True) (model/modeling_xllmx_dimoo.py:41-72) -> logits [B, L, vocab].

        use_cache / to_compute_mask / cat are LLaDAModelLM.forward's dLLM-cache arguments (model/modeling_llada.py:
        1468-1493,1244-1245,929-940,1406-1413): with use_cache=True the call goes through the cache slot of `cat`
        (forward_cached); to_compute_mask [B, L] bool then selects the tokens that are recomputed — every other position's
        keys, values and logits are reused — and the returned logits are the whole logit cache, as in the reference."""
        if not infer or labels is not None:
            raise NotImplementedError("only forward(infer=True) is on the MI355X hot path (training loss is out of scope)")
        if to_compute_mask is not None and not use_cache:
            raise ValueError("to_compute_mask needs use_cache=True (the reference only gathers the tokens then, "
                             "model/modeling_llada.py:1244-1245)")
        if use_cache and (self.tp_size == 1 or self._comm_in_library):
            # also under tensor parallelism when the exchange runs in the library (init_tp_comm): every rank caches its heads
            self.forward_cached(input_ids, to_compute_mask=to_compute_mask, cat=cat)
            B, L = self._cache[cat].shape
            rows = torch.arange(B * L, dtype=torch.int32, device=self.device)
            return CausalLMOutputLite(logits=self.cache_head_rows(cat, rows, 0, self.vocab).view(B, L, self.vocab))
        if to_compute_mask is not None:
            raise NotImplementedError("the dLLM cache under tensor parallelism needs the library's exchange (init_tp_comm); "
                                      "the host all-reduce fallback recomputes every row")
        # host all-reduce fallback + use_cache without a mask: every row is recomputed and nothing is kept (same logits)
        self.forward_body(input_ids)
        B, L = self._shape
        rows = torch.arange(B * L, dtype=torch.int32, device=self.device)
        logits = self.head_rows(rows, 0, self.vocab).view(B, L, self.vocab)
        return CausalLMOutputLite(logits=logits)

    __call__ = forward

    # ---- tensor-parallel transport (csrc/tp_comm.hip) ---------------------------------------------------------------
    class _DevView:
        """Zero-copy torch view of library-owned device memory (__cuda_array_interface__)."""

        def __init__(self, ptr, n_u16):
            self.__cuda_array_interface__ = {"shape": (n_u16,), "typestr": "<u2", "data": (ptr, False), "version": 2}

    def _part_view(self, rows: int) -> torch.Tensor:
        ptr = self._lib.mmada_comm_part_ptr(self._handle)
        n = rows * self.config.d_model
        return torch.as_tensor(self._DevView(ptr, n), device=self.device).view(torch.bfloat16).view(rows, self.config.d_model)

    def comm_status(self):
        mode, err, fine = C.c_int(), C.c_int(), C.c_int()
        abi.check(self._lib.mmada_comm_status(self._handle, C.byref(mode), C.byref(err), C.byref(fine), abi.stream_ptr()),
                  "mmada_comm_status")
        return {"mode": {0: "none", 1: "pull", 2: "rccl", 3: "no-exchange diagnostic", 4: "copy"}[mode.value], "error": err.value, "finegrained_counters": bool(fine.value & 1),
                "finegrained_buffers": bool(fine.value & 2)}

    def comm_selftest(self, iters: int = 3, L: int = 96) -> bool:
        """`iters` exchanges over a small carve with known partials (different data every round, so a stale cache line
        cannot pass): every row of the all-gathered, normalised result must equal the locally computed expectation bit for
        bit.  Every rank must call it; returns this rank's verdict."""
        d, tp, r = self.config.d_model, self.tp_size, self.tp_rank
        B = 2
        ids = (torch.arange(B * L, device=self.device).view(B, L) * 7 + 3) % 1000
        Lp = (L + 7) // 8 * 8
        M = B * Lp
        self._ensure_ws(B, L)
        w = torch.ones(d, dtype=torch.bfloat16, device=self.device)
        part = self._part_view(M)
        st = abi.stream_ptr()
        ok = True
        col = torch.arange(d, device=self.device, dtype=torch.float32)[None, :]
        row = torch.arange(M, device=self.device, dtype=torch.float32)[:, None]
        for it in range(iters):
            abi.check(self._lib.mmada_embed(self._handle, ids.data_ptr(), B, L, st), "mmada_embed")
            self._shape, self._split = (B, L), None
            x0 = self._stream_view().view(M, d).clone()

            def pat(rank):  # small integers: exact in bf16, different per rank / row / column / round
                return (((row * 3 + col * 5 + rank * 11 + it * 17) % 13) - 6.0) * (rank + 1)

            part.copy_(pat(r).to(torch.bfloat16))
            abi.check(self._lib.mmada_comm_exchange(self._handle, w.data_ptr(), st), "mmada_comm_exchange")
            total = sum(pat(j).to(torch.bfloat16).float() for j in range(tp))
            x_new = (x0.float() + total.to(torch.bfloat16).float()).to(torch.bfloat16)
            want = torch.empty_like(x_new)
            abi.check(self._lib.mmada_rmsnorm(x_new.data_ptr(), w.data_ptr(), want.data_ptr(), M, d,
                                              float(self.config.ref("rms_norm_eps")), st), "mmada_rmsnorm")
            got = self.debug_buffer(0).view(-1, d)[:M]
            ok = ok and bool(torch.equal(got, want))
        return ok and self.comm_status()["error"] == 0

    def init_tp_comm(self, max_batch: int, max_len: int, group=None, transport: str = "auto") -> str:
        """Connect the library's tensor-parallel exchange over the ranks of `group` (a torch.distributed group: control
        plane only — handles / unique id are exchanged as objects; the data path never goes through torch).
        transport: "pull" (mapped peer buffers, hipIpc), "copy" (the same mapped buffers, bytes moved by the copy engines),
        "rccl", or "auto" = pull if it connects AND passes the self-test on every rank, else RCCL, else the host-issued
        all-reduce of the segment API.  Returns what is in use.  MMADA_TP_EXCHANGE_CUS=n (a multiple of 8) additionally gives
        the exchange stream n CUs of its own and masks the compute stream to the rest (mmada_comm_set_partition)."""
        import torch.distributed as dist

        lib = self._lib
        rows = max_batch * ((max_len + 7) // 8 * 8)
        nb = lib.mmada_comm_export_bytes()
        buf = C.create_string_buffer(nb)
        exported = lib.mmada_comm_create(self._handle, rows, buf) == 0
        if not exported:
            abi.check(lib.mmada_comm_create(self._handle, rows, None), "mmada_comm_create")
        self._comm_rows = rows

        def all_agree(flag: bool) -> bool:
            got = [None] * self.tp_size
            dist.all_gather_object(got, bool(flag), group=group)
            return all(got)

        chosen = None
        if transport in ("auto", "pull", "copy"):
            blobs = [None] * self.tp_size
            dist.all_gather_object(blobs, buf.raw if exported else None, group=group)
            ok = exported and all(b is not None for b in blobs)
            if ok:
                ok = lib.mmada_comm_connect_ipc(self._handle, b"".join(blobs)) == 0
            ok = all_agree(ok)
            if ok and transport == "copy":
                ok = all_agree(lib.mmada_comm_set_mode(self._handle, 4) == 0)
            if ok:
                self._comm_in_library = True
                lib.mmada_comm_set_timeout(self._handle, 3.0)   # a transport that cannot work is abandoned quickly
                ok = all_agree(self.comm_selftest())
                lib.mmada_comm_set_timeout(self._handle, 0.0)   # back to MMADA_TP_TIMEOUT_S; clears a sticky error
            if ok:
                chosen = "copy" if transport == "copy" else "pull"
            elif transport in ("pull", "copy"):
                raise abi.MmadaError("tensor-parallel pull transport failed to connect or failed its self-test: "
                                     + (lib.mmada_last_error() or b"").decode())
        if chosen is None and transport in ("auto", "rccl"):
            backend_ok = dist.get_backend(group) == "nccl"  # one rank per device: RCCL refuses two ranks on one GPU
            if all_agree(backend_ok):
                path = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so").encode()
                uid = C.create_string_buffer(128)
                if self.tp_rank == 0:
                    abi.check(lib.mmada_comm_unique_id(uid, path), "mmada_comm_unique_id")
                box = [uid.raw]
                dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
                ok = lib.mmada_comm_connect_rccl(self._handle, box[0], path) == 0
                if all_agree(ok):
                    self._comm_in_library = True
                    if all_agree(self.comm_selftest()):
                        chosen = "rccl"
            if chosen is None and transport == "rccl":
                raise abi.MmadaError("tensor-parallel RCCL transport failed: " + (lib.mmada_last_error() or b"").decode())
        if chosen is None:
            self._comm_in_library = False
            lib.mmada_comm_destroy(self._handle)
            chosen = "host all-reduce (torch.distributed)"
        self._rccl_also = False
        if chosen in ("pull", "copy") and os.environ.get("MMADA_TP_PROBE_RCCL", "0") == "1":
            # OPT-IN (bench.py sets it): one rank per device over RCCL as well, so that collective_probe() can time BOTH
            # transports.  A production start does not pay a second communicator (init time, memory, one more thing that
            # can fail or hang at start-up).
            if all_agree(dist.get_backend(group) == "nccl"):
                path = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so").encode()
                uid = C.create_string_buffer(128)
                uid_ok = self.tp_rank != 0 or lib.mmada_comm_unique_id(uid, path) == 0
                box = [uid.raw if uid_ok else None]
                dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
                if box[0] is not None:
                    try:
                        ok = lib.mmada_comm_connect_rccl(self._handle, box[0], path) == 0   # leaves mode = RCCL
                    finally:
                        lib.mmada_comm_set_mode(self._handle, 4 if chosen == "copy" else 1)   # the forward keeps its transport
                    self._rccl_also = all_agree(ok)
        self.tp_collective = chosen
        cus = int(os.environ.get("MMADA_TP_EXCHANGE_CUS", "0") or 0)
        if cus and self._comm_in_library:
            abi.check(lib.mmada_comm_set_partition(self._handle, cus), "mmada_comm_set_partition")
        return chosen

    def set_exchange_partition(self, exchange_cus: int) -> None:
        """Give the exchange stream `exchange_cus` CUs of its own (0: none) — mmada_comm_set_partition."""
        abi.check(self._lib.mmada_comm_set_partition(self._handle, int(exchange_cus)), "mmada_comm_set_partition")

    def collective_probe(self, L: int, B: int = 1, iters: int = 10):
        """Outside any timed region: one exchange (reduce-scatter + RMSNorm + all-gather of B*L rows x d bf16) timed alone,
        so a scaling run also records what the fabric delivered for the message size the forward uses."""
        if not self._comm_in_library:
            return None
        import time

        ids = torch.zeros((B, L), dtype=torch.long, device=self.device)
        self._ensure_ws(B, L)
        st = abi.stream_ptr()
        abi.check(self._lib.mmada_embed(self._handle, ids.data_ptr(), B, L, st), "mmada_embed")
        w = torch.ones(self.config.d_model, dtype=torch.bfloat16, device=self.device)
        for _ in range(3):
            abi.check(self._lib.mmada_comm_exchange(self._handle, w.data_ptr(), st), "mmada_comm_exchange")
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            abi.check(self._lib.mmada_comm_exchange(self._handle, w.data_ptr(), st), "mmada_comm_exchange")
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / iters * 1e3
        nbytes = B * ((L + 7) // 8 * 8) * self.config.d_model * 2
        tp = self.tp_size
        out = {"transport": self.tp_collective, "rows": B * L, "bytes": nbytes, "ms": ms,
               "busbw_GBps": 2.0 * (tp - 1) / tp * nbytes / (ms * 1e-3) / 1e9, "exchanges_per_forward": 2 * self.config.n_layers,
               "status": self.comm_status()}
        # the other data path over the same mapped buffers, for comparison: OPT-IN (MMADA_TP_PROBE_COPY=1) — it exercises a
        # transport the run did not select; a first multi-GPU session should ask for it explicitly
        if self.tp_collective in ("pull", "copy") and os.environ.get("MMADA_TP_PROBE_COPY", "0") == "1":
            other, mode_other, mode_back = ("copy", 4, 1) if self.tp_collective == "pull" else ("pull", 1, 4)
            # a comparison only: a data path that fails HERE (first contact with real multi-GPU hardware) must not take the
            # benchmark line of the transport in use with it — record the error and go on
            err_in_use = out["status"]["error"]   # what the transport IN USE left behind: recorded above, never erased below
            ok = 1
            try:
                abi.check(self._lib.mmada_comm_set_mode(self._handle, mode_other), "mmada_comm_set_mode")
                for _ in range(3):
                    abi.check(self._lib.mmada_comm_exchange(self._handle, w.data_ptr(), st), "mmada_comm_exchange")
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(iters):
                    abi.check(self._lib.mmada_comm_exchange(self._handle, w.data_ptr(), st), "mmada_comm_exchange")
                torch.cuda.synchronize()
                ms3 = (time.perf_counter() - t0) / iters * 1e3
                out[other] = {"ms": ms3, "busbw_GBps": 2.0 * (tp - 1) / tp * nbytes / (ms3 * 1e-3) / 1e9,
                              "error_flag": self.comm_status()["error"]}
                ok = int(out[other]["error_flag"] == 0)
            except Exception as e:   # noqa: BLE001
                out[other] = {"error": str(e)[:300]}
                ok = 0
            finally:
                abi.check(self._lib.mmada_comm_set_mode(self._handle, mode_back), "mmada_comm_set_mode")
                if err_in_use == 0:   # only a flag the COMPARISON raised is cleared; an earlier one stays for bench.py to report
                    self._lib.mmada_comm_set_timeout(self._handle, 0.0)
            # a rank that failed stopped issuing exchanges while its peers went on: agree on the outcome before anything else
            # uses the group (the comparison's figure is only meaningful when every rank completed it)
            import torch.distributed as dist
            if dist.is_available() and dist.is_initialized():
                flag = torch.tensor([ok], dtype=torch.int32, device=self.device)
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                out[other]["all_ranks_ok"] = bool(int(flag.item()))
        if getattr(self, "_rccl_also", False) and self.tp_collective in ("pull", "copy"):   # the same exchange over RCCL, for comparison
            abi.check(self._lib.mmada_comm_set_mode(self._handle, 2), "mmada_comm_set_mode")
            try:
                for _ in range(3):
                    abi.check(self._lib.mmada_comm_exchange(self._handle, w.data_ptr(), st), "mmada_comm_exchange")
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(iters):
                    abi.check(self._lib.mmada_comm_exchange(self._handle, w.data_ptr(), st), "mmada_comm_exchange")
                torch.cuda.synchronize()
                ms2 = (time.perf_counter() - t0) / iters * 1e3
                out["rccl"] = {"ms": ms2, "busbw_GBps": 2.0 * (tp - 1) / tp * nbytes / (ms2 * 1e-3) / 1e9}
            finally:
                abi.check(self._lib.mmada_comm_set_mode(self._handle, 4 if self.tp_collective == "copy" else 1), "mmada_comm_set_mode")
        return out

    def rccl_nranks(self) -> int:
        """Ranks of the RCCL communicator the LIBRARY created (ncclCommCount), 0 when it holds none."""
        return int(self._lib.mmada_comm_rccl_nranks(self._handle)) if self._comm_in_library or getattr(self, "_rccl_also", False) else 0

    def exchange_exposure_probe(self, input_ids: torch.Tensor, reps: int = 3):
        """Outside any timed region: wall time of one tensor-parallel forward with its exchanges and of the same forward
        with the library's "no exchange" diagnostic (mmada_comm_set_mode 3: identical GEMM / attention / owner-side kernels,
        no peer traffic, no hand-off; the values are wrong, only the time is used).  The difference is what the exchanges
        cost the forward AFTER the two-chunk overlap: the exposed exchange time.  Every rank must call it."""
        if not self._comm_in_library or self.tp_size == 1:
            return None
        import time

        import torch.distributed as dist

        real_mode = {"pull": 1, "rccl": 2, "copy": 4}[self.tp_collective]

        def timed(mode):
            abi.check(self._lib.mmada_comm_set_mode(self._handle, mode), "mmada_comm_set_mode")
            ts = []
            try:
                for i in range(reps + 1):
                    if dist.is_initialized():
                        dist.barrier()
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    self.forward_body(input_ids)
                    torch.cuda.synchronize()
                    if i:   # the first call of a mode is a warm-up
                        ts.append((time.perf_counter() - t0) * 1e3)
            finally:
                abi.check(self._lib.mmada_comm_set_mode(self._handle, real_mode), "mmada_comm_set_mode")
            return sorted(ts)[len(ts) // 2]

        with_x = timed(real_mode)
        without = timed(3)
        with_x2 = timed(real_mode)
        ms = min(with_x, with_x2)
        return {"forward_ms_with_exchange": ms, "forward_ms_no_exchange_diagnostic": without,
                "exposed_exchange_ms_per_forward": ms - without, "exchanges_per_forward": 2 * self.config.n_layers,
                "batch": int(input_ids.shape[0]), "what": "median wall time of a synchronised forward_body, real transport vs "
                "mmada_comm_set_mode(3) (owner-side kernels on the rank's own partials only, no peer traffic)"}

    def vocab_parallel_head(self) -> bool:
        """True when the text step can run on vocabulary slices of the LM head (library transport connected)."""
        return self._comm_in_library and os.environ.get("MMADA_TP_REPLICATED_HEAD") != "1"

    def graph_capturable(self) -> bool:
        """True when forward_body / head_rows issue only stream launches (no host-side collective): the sampler may then
        capture a whole denoise step into one hipGraph (mmada_graph_*)."""
        # RCCL's reduce-scatter / all-gather would be captured on a forked stream; whether every call it makes is
        # capturable has never been exercised with more than one rank, so only the pull transport (plain kernels and
        # device-memory counters) qualifies under tensor parallelism
        return self.tp_size == 1 or (getattr(self, "_comm_in_library", False) and getattr(self, "tp_collective", None) in ("pull", "copy")
                                     and self._lib.mmada_comm_partition(self._handle) == 0)   # a CU mask does not survive capture

    # ---- dLLM cache (model/modeling_llada.py:593-600,929-940,1244-1245,1406-1426) ----------------------------------------
    def caching(self, enable: bool = True):
        """LLaDAModel.caching (model/modeling_llada.py:1417-1421 -> every block's :598-600): sets the blocks' use_cache flag
        — which only decides whether the queries of a compute-mask step are rotated by their own positions (:714-716) — and
        clears every cache."""
        self.use_cache = bool(enable)
        self.empty_cache()

    def empty_cache(self):
        """LLaDAModel.empty_cache (model/modeling_llada.py:1423-1426): drops every slot's keys / values / final rows."""
        for ent in getattr(self, "_cache", {}).values():
            abi.check(self._lib.mmada_cache_bind(self._handle, ent.idx, None, 0, 0, 0, abi.stream_ptr()), "mmada_cache_bind")
        self._cache = {}

    def _cache_slot(self, cat, B: int, L: int, rebind_ok: bool):
        from types import SimpleNamespace

        cache = self.__dict__.setdefault("_cache", {})
        ent = cache.get(cat)
        if ent is not None and ent.shape == (B, L):
            return ent
        if ent is not None and not rebind_ok:
            raise ValueError(f"cache {cat!r} holds sequences of shape {ent.shape}, the masked call has {(B, L)}")
        if ent is None:
            used = {e.idx for e in cache.values()}
            free = [i for i in range(16) if i not in used]
            if not free:
                raise abi.MmadaError("all 16 dLLM cache slots are in use (empty_cache() releases them)")
            idx = free[0]
        else:
            idx = ent.idx
        nbytes = self._lib.mmada_cache_bytes(self._handle, B, L)
        mem = torch.empty(nbytes + 256, dtype=torch.uint8, device=self.device)
        base = (mem.data_ptr() + 255) // 256 * 256
        abi.check(self._lib.mmada_cache_bind(self._handle, idx, base, nbytes, B, L, abi.stream_ptr()), "mmada_cache_bind")
        cache[cat] = SimpleNamespace(idx=idx, mem=mem, shape=(B, L))
        return cache[cat]

    def forward_cached(self, input_ids: torch.Tensor, to_compute_mask: Optional[torch.Tensor] = None, cat="") -> None:
        """One forward through the cache slot `cat` (created at zeros on first use, like the reference's zeros_like).
        Mask None: every token is computed and the slot is (re)filled.  Mask [B, L] bool with the same count in every
        row (the reference's `.view(B, -1)`): only those tokens run through the blocks; read logits with cache_head_rows."""
        if self.tp_size != 1 and not self._comm_in_library:
            raise NotImplementedError("the dLLM cache path under tensor parallelism needs the library's exchange (init_tp_comm)")
        ids = input_ids.to(device=self.device, dtype=torch.long).contiguous()
        B, L = ids.shape
        if self._comm_in_library and B * ((L + 7) // 8 * 8) > self._comm_rows:
            raise abi.MmadaError(f"cached forward of {B}x{L} exceeds the {self._comm_rows} rows init_tp_comm() was sized for")
        self._ensure_ws(B, L)
        ent = self._cache_slot(cat, B, L, rebind_ok=to_compute_mask is None)
        st = abi.stream_ptr()
        if to_compute_mask is None:
            abi.check(self._lib.mmada_forward_cached(self._handle, ent.idx, ids.data_ptr(), None, B, L, L, 1, st),
                      "mmada_forward_cached")
        else:
            m = to_compute_mask.to(device=self.device, dtype=torch.bool)
            if m.shape != (B, L):
                raise ValueError(f"to_compute_mask {tuple(m.shape)} does not match input_ids {(B, L)}")
            cnt = m.sum(1)
            Tc = int(cnt[0])
            if Tc == 0 or not bool((cnt == Tc).all()):
                raise ValueError("to_compute_mask must select the same, non-zero number of tokens in every sequence")
            pos = m.nonzero()[:, 1].view(B, Tc).to(torch.int32).contiguous()
            ids_c = ids[m].view(B, Tc).contiguous()
            abi.check(self._lib.mmada_forward_cached(self._handle, ent.idx, ids_c.data_ptr(), pos.data_ptr(), B, L, Tc,
                                                     int(bool(getattr(self, "use_cache", False))), st), "mmada_forward_cached")
        self._shape, self._split, self._consumed = None, None, None  # no plain forward is resident any more

    def cache_head_rows(self, cat, rows: torch.Tensor, col_begin: int, col_end: int,
                        out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Rows (b*L + l) x columns [col_begin, col_end) of logit_cache[cat] (model/modeling_llada.py:1406-1413)."""
        ent = getattr(self, "_cache", {}).get(cat)
        if ent is None:
            raise KeyError(f"no cache {cat!r}")
        rows = rows.to(device=self.device, dtype=torch.int32).contiguous()
        if out is None:
            out = torch.empty((rows.numel(), col_end - col_begin), dtype=torch.bfloat16, device=self.device)
        elif out.shape != (rows.numel(), col_end - col_begin) or out.dtype != torch.bfloat16 or not out.is_contiguous():
            raise ValueError("cache_head_rows: `out` must be a contiguous bf16 [R, col_end-col_begin] tensor")
        abi.check(self._lib.mmada_cache_head_rows(self._handle, ent.idx, rows.data_ptr(), rows.numel(), col_begin, col_end,
                                                  out.data_ptr(), abi.stream_ptr()), "mmada_cache_head_rows")
        return out

    def eval(self):
        return self

    def to(self, *_a, **_k):
        return self

    def __del__(self):
        try:
            if getattr(self, "_handle1", None) is not None and self._handle1.value:
                self._lib.mmada_destroy(self._handle1)
                self._handle1 = None
            if getattr(self, "_handle", None) and self._handle.value:
                self._lib.mmada_destroy(self._handle)
                self._handle = C.c_void_p()
        except Exception:
            pass
