"""`ldm.modules.diffusionmodules.openaimodel` on the MI355X kernels.

Same public surface as the reference (ldm/modules/diffusionmodules/openaimodel.py): TimestepBlock 60-70,
TimestepEmbedSequential 73-87, Upsample 90-118, Downsample 133-159, ResBlock 162-274, UNetModel 412-787 -- identical
constructor kwargs, forward signatures and the 686 state-dict keys of the SD2-inpainting UNet, so checkpoints load
unchanged.  Internally the network runs on NHWC fp16 activations:

  * the whole forward is a flat plan of HIP kernel launches (leftrefill_amd.engine) -- implicit-GEMM convs on MFMA
    with the time-embedding add / residual add fused in the epilogue, GroupNorm+SiLU streaming kernels reading the
    skip concat virtually, fused transformer blocks, one batched GEMV for the 22 emb_layers projections;
  * shapes are static across the 50 DDIM steps, so the plan is captured once per input shape into a hipGraph
    (torch.cuda.CUDAGraph on ROCm) and replayed: one graph launch per UNet step instead of ~450 kernel launches.

The product path has no PyTorch-eager fallback; a missing HIP library raises.
"""
from abc import abstractmethod

import os

import torch
import torch as th
import torch.nn as nn

from leftrefill_amd import engine
from leftrefill_amd import train_ops as ops     # == leftrefill_amd.ops unless autograd records an input that requires grad
from ldm.modules.attention import SpatialTransformer
from ldm.modules.diffusionmodules.util import conv_nd, linear, normalization, zero_module

# training: every cross-attention's [k;v] projection of the context as ONE GEMM (and one input-gradient GEMM); 0 = one per block
TRAIN_KV_BATCH = os.environ.get("LEFTREFILL_TRAIN_KV_BATCH", "1") != "0"
# a sampler's per-timestep embedding rows computed once per sampling (UNetModel.prepare_timesteps); 0 = recompute them every step
EMB_TABLE = os.environ.get("LEFTREFILL_EMB_TABLE", "1") != "0"


class TimestepBlock(nn.Module):
    """Any module whose forward takes the timestep embedding as second argument."""

    @abstractmethod
    def forward(self, x, emb):
        ...


class TimestepEmbedSequential(nn.Sequential, TimestepBlock):
    """Routes `emb` to TimestepBlocks and `context` to SpatialTransformers (reference 73-87)."""

    def forward(self, x, emb, context=None, **kwargs):
        for layer in self:
            if isinstance(layer, TimestepBlock):
                x = layer(x, emb)
            elif isinstance(layer, SpatialTransformer):
                x = layer(x, context)
            else:
                x = layer(x)
        return x


class Upsample(nn.Module):
    """Nearest x2 followed by conv3x3: the upsample is index arithmetic inside the conv's gather (no copy)."""

    def __init__(self, channels, use_conv, dims=2, out_channels=None, padding=1):
        super().__init__()
        self.channels = channels
        self.out_channels = out_channels or channels
        self.use_conv = use_conv
        self.dims = dims
        if not use_conv or dims != 2 or padding != 1:
            raise NotImplementedError("only the learned 2-D upsample (conv_resample=True) is used by LeftRefill")
        self.conv = conv_nd(dims, self.channels, self.out_channels, 3, padding=padding)

    def _fwd(self, act):
        pc = engine_cache(self, "conv", lambda: engine.PackedConv(self.conv))
        return engine.conv(act, pc, up=1)

    def forward(self, x):
        assert x.shape[1] == self.channels
        return engine.act_to_nchw(self._fwd(engine.act_from_nchw(x)), self.out_channels).to(x.dtype)


class Downsample(nn.Module):
    """conv3x3 stride 2, symmetric pad 1 (reference 150-152)."""

    def __init__(self, channels, use_conv, dims=2, out_channels=None, padding=1):
        super().__init__()
        self.channels = channels
        self.out_channels = out_channels or channels
        self.use_conv = use_conv
        self.dims = dims
        if not use_conv or dims != 2 or padding != 1:
            raise NotImplementedError("only the learned 2-D downsample (conv_resample=True) is used by LeftRefill")
        self.op = conv_nd(dims, self.channels, self.out_channels, 3, stride=2, padding=padding)

    def _fwd(self, act):
        pc = engine_cache(self, "op", lambda: engine.PackedConv(self.op))
        return engine.conv(act, pc)

    def forward(self, x):
        assert x.shape[1] == self.channels
        return engine.act_to_nchw(self._fwd(engine.act_from_nchw(x)), self.out_channels).to(x.dtype)


def engine_cache(module, key, build):
    from ldm.modules.attention import _cached
    return _cached(module, key, build)


class ResBlock(TimestepBlock):
    """GN+SiLU -> conv3x3 (+emb) -> GN+SiLU -> conv3x3 -> + skip(x)   (reference 254-274)."""

    def __init__(self, channels, emb_channels, dropout, out_channels=None, use_conv=False, use_scale_shift_norm=False,
                 dims=2, use_checkpoint=False, up=False, down=False):
        super().__init__()
        if use_scale_shift_norm or up or down or use_conv:
            raise NotImplementedError("scale-shift norm / resblock_updown / conv skip are not used by LeftRefill")
        self.channels = channels
        self.emb_channels = emb_channels
        self.dropout = dropout
        self.out_channels = out_channels or channels
        self.use_conv = use_conv
        self.use_checkpoint = use_checkpoint
        self.use_scale_shift_norm = use_scale_shift_norm
        self.updown = False
        self.in_layers = nn.Sequential(normalization(channels), nn.SiLU(),
                                       conv_nd(dims, channels, self.out_channels, 3, padding=1))
        self.h_upd = self.x_upd = nn.Identity()
        self.emb_layers = nn.Sequential(nn.SiLU(), linear(emb_channels, self.out_channels))
        self.out_layers = nn.Sequential(normalization(self.out_channels), nn.SiLU(), nn.Dropout(p=dropout),
                                        zero_module(conv_nd(dims, self.out_channels, self.out_channels, 3, padding=1)))
        if self.out_channels == channels:
            self.skip_connection = nn.Identity()
        else:
            self.skip_connection = conv_nd(dims, channels, self.out_channels, 1)

    def _packed(self):
        return engine_cache(self, "res", lambda: engine.PackedRes(self))

    def _fwd(self, act, emb_out):
        return engine.resblock(act, self._packed(), emb_out)

    def forward(self, x, emb):
        pr = self._packed()
        e = ops.linear_small_m(emb.to(pr.emb.w.dtype).contiguous(), pr.emb.w, pr.emb.b, act_in=True)
        out = self._fwd(engine.act_from_nchw(x), e)
        return engine.act_to_nchw(out, self.out_channels).to(x.dtype)


class UNetModel(nn.Module):
    """SD UNet with spatial transformers; constructor kwargs as in the reference (openaimodel.py:442-472)."""

    st_cls = SpatialTransformer
    st_kwargs = {}

    def __init__(self, image_size, in_channels, model_channels, out_channels, num_res_blocks, attention_resolutions,
                 dropout=0, channel_mult=(1, 2, 4, 8), conv_resample=True, dims=2, num_classes=None,
                 use_checkpoint=False, use_fp16=False, num_heads=-1, num_head_channels=-1, num_heads_upsample=-1,
                 use_scale_shift_norm=False, resblock_updown=False, use_new_attention_order=False,
                 use_spatial_transformer=False, transformer_depth=1, context_dim=None, n_embed=None, legacy=True,
                 disable_self_attentions=None, num_attention_blocks=None, disable_middle_self_attn=False,
                 use_linear_in_transformer=False):
        super().__init__()
        if not use_spatial_transformer or context_dim is None:
            raise NotImplementedError("LeftRefill's UNet always uses the spatial transformer with a context_dim")
        if num_classes is not None or n_embed is not None or resblock_updown or use_scale_shift_norm or dims != 2:
            raise NotImplementedError("class-conditional / codebook / resblock_updown variants are out of scope")
        if disable_self_attentions is not None or num_attention_blocks is not None or disable_middle_self_attn:
            raise NotImplementedError("disable_self_attentions / num_attention_blocks are unused by LeftRefill")
        if not isinstance(context_dim, int):
            context_dim = list(context_dim)
            assert len(context_dim) == 1 or transformer_depth == len(context_dim)
        if num_heads == -1 and num_head_channels == -1:
            raise AssertionError("Either num_heads or num_head_channels has to be set")
        if num_heads_upsample == -1:
            num_heads_upsample = num_heads
        self.image_size = image_size
        self.in_channels = in_channels
        self.model_channels = model_channels
        self.out_channels = out_channels
        self.transformer_depth = transformer_depth
        self.num_res_blocks = len(channel_mult) * [num_res_blocks] if isinstance(num_res_blocks, int) \
            else list(num_res_blocks)
        if len(self.num_res_blocks) != len(channel_mult):
            raise ValueError("provide num_res_blocks either as an int or as a per-level list")
        self.attention_resolutions = attention_resolutions
        self.dropout = dropout
        self.channel_mult = channel_mult
        self.conv_resample = conv_resample
        self.num_classes = num_classes
        self.use_checkpoint = use_checkpoint
        self.dtype = th.float16 if use_fp16 else th.float32
        # 16-bit type of the HIP path's activations / packed weights: float16 (reference autocast, appendix B) or bfloat16
        self.compute_dtype = th.float16
        # Set by DDIMSampler while it runs classifier-free guidance on an [x; x] batch (the reference's
        # `torch.cat([x] * 2)`, ddim.py:317-342): both halves see the same latent, timestep and concat conditioning, and the
        # context enters only at the first cross-attention, so conv_in, the first ResBlock, the first self-attention and
        # the first cross-attention's query projection are computed ONCE for the two halves and duplicated.  Exact (no
        # approximation); `LEFTREFILL_CFG_SHARED_PREFIX=0` keeps the sampler from setting it.
        self.cfg_shared_prefix = False
        self.num_heads = num_heads
        self.num_head_channels = num_head_channels
        self.num_heads_upsample = num_heads_upsample
        self.predict_codebook_ids = False

        mc = model_channels
        ted = mc * 4
        self.time_embed = nn.Sequential(linear(mc, ted), nn.SiLU(), linear(ted, ted))

        def heads_for(ch):
            if num_head_channels == -1:
                nh = num_heads
                dh = ch // nh
            else:
                nh = ch // num_head_channels
                dh = num_head_channels
            if legacy:
                dh = ch // nh
            return nh, dh

        def make_st(ch):
            nh, dh = heads_for(ch)
            return self.st_cls(ch, nh, dh, depth=transformer_depth, context_dim=context_dim,
                               use_linear=use_linear_in_transformer, use_checkpoint=use_checkpoint, **self.st_kwargs)

        def make_res(cin, cout):
            return ResBlock(cin, ted, dropout, out_channels=cout, dims=dims, use_checkpoint=use_checkpoint)

        self.input_blocks = nn.ModuleList([TimestepEmbedSequential(conv_nd(dims, in_channels, mc, 3, padding=1))])
        skip_chans = [mc]
        ch, ds = mc, 1
        for level, mult in enumerate(channel_mult):
            for _ in range(self.num_res_blocks[level]):
                layers = [make_res(ch, mult * mc)]
                ch = mult * mc
                if ds in attention_resolutions:
                    layers.append(make_st(ch))
                self.input_blocks.append(TimestepEmbedSequential(*layers))
                skip_chans.append(ch)
            if level != len(channel_mult) - 1:
                self.input_blocks.append(TimestepEmbedSequential(Downsample(ch, conv_resample, dims=dims,
                                                                            out_channels=ch)))
                skip_chans.append(ch)
                ds *= 2
        self.middle_block = TimestepEmbedSequential(make_res(ch, ch), make_st(ch), make_res(ch, ch))
        self.output_blocks = nn.ModuleList([])
        for level, mult in list(enumerate(channel_mult))[::-1]:
            for i in range(self.num_res_blocks[level] + 1):
                layers = [make_res(ch + skip_chans.pop(), mc * mult)]
                ch = mc * mult
                if ds in attention_resolutions:
                    layers.append(make_st(ch))
                if level and i == self.num_res_blocks[level]:
                    layers.append(Upsample(ch, conv_resample, dims=dims, out_channels=ch))
                    ds //= 2
                self.output_blocks.append(TimestepEmbedSequential(*layers))
        self.out = nn.Sequential(normalization(ch), nn.SiLU(),
                                 zero_module(conv_nd(dims, mc, out_channels, 3, padding=1)))
        # engine state
        self.recompute_in_backward = False
        self.use_hip_graph = True
        self._plan = None
        self._graphs = {}

    # ------------------------------------------------------------------------------------------------------
    # weight preparation: reference-shaped nn.Parameters -> packed fp16 kernel layouts (once, after loading)
    # ------------------------------------------------------------------------------------------------------
    MAX_GRAPHS = 8        # captured step graphs kept per model (least recently used shape is dropped beyond that)

    def _sig(self):
        return (self.compute_dtype,) + tuple((p.data_ptr(), p._version) for p in self.parameters())

    def _load_from_state_dict(self, *args, **kwargs):
        super()._load_from_state_dict(*args, **kwargs)
        self._weights_dirty = True          # load_state_dict / copy into parameters: re-pack on the next forward

    def invalidate_context_cache(self):
        """Forget the cached cross-attention K/V projections of every captured step graph.

        Contract of the cache: a context tensor is recognised by object identity + autograd `_version`.  Writes that bypass
        version counting (`ctx.data.copy_`, raw-pointer kernels writing into it, e.g. `ops.* (out=ctx)`) are NOT seen --
        call this after such a write (or pass a new tensor)."""
        for g in self._graphs.values():
            g.ctx_src = None

    def prepare(self, force=False):
        """Pack all weights for the kernels.  Re-packs when forced, after load_state_dict, or when a parameter was replaced /
        modified in place through autograd-visible ops; `.data` writes need `prepare(force=True)`."""
        sig = self._sig()      # ~0.3 ms of host time per call; the captured step keeps the GPU busy for ~20 ms
        if self._plan is not None and not force and not getattr(self, "_weights_dirty", False) and self._plan["sig"] == sig:
            return self._plan
        self._weights_dirty = False
        dev = next(self.parameters()).device
        if dev.type != "cuda":
            raise RuntimeError("UNetModel runs on the HIP device only: call .to('cuda') first (no CPU fallback)")
        with engine.compute(self.compute_dtype):
            return self._pack(sig)

    def _pack(self, sig):
        E = engine
        plan = {"sig": sig}
        plan["t0"] = E.PackedLinear(self.time_embed[0])
        plan["t2"] = E.PackedLinear(self.time_embed[2])
        # the 9 input channels travel as 16 (32 bytes per pixel, lr_gemm_conv_f16's 16-channel gather); wider inputs pad to 64s
        # (not for the separator-token variant, NVSUnetModel.use_sep: its input conv needs an input gradient, and the dgrad weights are
        # derived from the [N][taps * C] layout)
        c16 = self.in_channels <= 16 and not getattr(self, "use_sep", False) and os.environ.get("LEFTREFILL_CONV_IN_C16", "1") != "0"
        cin_pad = 16 if c16 else max(64, ((self.in_channels + 63) // 64) * 64)
        plan["cin_pad"] = cin_pad
        res_list = []

        def pack_seq(seq):
            steps = []
            for layer in seq:
                if isinstance(layer, ResBlock):
                    pr = E.PackedRes(layer)
                    res_list.append(pr)
                    steps.append(("res", pr))
                elif isinstance(layer, SpatialTransformer):
                    steps.append(("st", E.PackedST(layer)))
                elif isinstance(layer, Downsample):
                    steps.append(("down", E.PackedConv(layer.op)))
                elif isinstance(layer, Upsample):
                    steps.append(("up", E.PackedConv(layer.conv)))
                elif isinstance(layer, nn.Conv2d):
                    steps.append(("conv", E.PackedConv(layer, cin_pad=cin_pad)))
                else:
                    raise TypeError(type(layer))
            return steps

        plan["input"] = [pack_seq(b) for b in self.input_blocks]
        plan["middle"] = pack_seq(self.middle_block)
        plan["output"] = [pack_seq(b) for b in self.output_blocks]
        plan["out_norm"] = E.PackedNorm(self.out[0])
        plan["out_conv"] = E.PackedConv(self.out[2])
        # all emb_layers projections as one [sum Cout, 4*mc] weight (one launch per step instead of 22)
        plan["emb_w"] = torch.cat([pr.emb.w for pr in res_list], 0).contiguous()
        plan["emb_b"] = torch.cat([pr.emb.b for pr in res_list], 0).contiguous()
        off = 0
        for pr in res_list:
            pr.emb_off = off
            off += pr.cout
        plan["emb_total"] = off
        # cross-attention K/V projections depend only on the context: one cache slot per transformer block
        tblocks = []
        for steps in plan["input"] + [plan["middle"]] + plan["output"]:
            for kind, p_ in steps:
                if kind == "st":
                    tblocks.extend(p_.blocks)
        for i, pt in enumerate(tblocks):
            pt.kv_slot = i
        plan["tblocks"] = tblocks
        self._plan = plan
        self._graphs = {}
        return plan

    # ------------------------------------------------------------------------------------------------------
    def _context_kv(self, context, out=None):
        """[k;v] projections of the context for every cross-attention (attention.py:171-172).  The context is the
        same tensor for all 50 DDIM steps, so the captured step graph reads these from a per-context cache."""
        P = self._plan
        N, L, D = context.shape
        ctx = context.reshape(N * L, D)
        res = []
        for i, pt in enumerate(P["tblocks"]):
            o, ot, xk, xvt = (None,) * 4 if out is None else (tuple(out[i]) + (None, None))[:4]
            kv = ops.gemm_conv(ctx, pt.attn2.kv.w, B=1, H=1, W=N * L, taps=1, out=o)
            C = kv.shape[1] // 2
            # V^T copy for lr_attention_vt_f16 (none when the kernel reads V with the LDS transpose read, ops.ATTN_VT = 0)
            ent = (kv, ops.transpose_v(kv[:, C:], N, pt.attn2.heads, L, out=ot) if ops.ATTN_VT else None)
            if pt.attn2.xk is not None and L <= ops.XATTN_MAX_KEYS:
                # operands of the fused cross-attention block: K in its k-slot order, V^T pack
                ent += (ops.gemm_conv(ctx, pt.attn2.xk, B=1, H=1, W=N * L, taps=1, out=xk),
                        ops.xattn_pack_vt(kv[:, C:], N, pt.attn2.heads, L, out=xvt))
            res.append(ent)
        return res

    # hooks of subclasses that reshape a block's input / output (inpainting_ldm.NVS_ldm.NVSUnetModel: separator column, c_input)
    def _block_in(self, act, steps):
        return act, None

    def _block_out(self, act, state):
        return act

    def _embed(self, timesteps):
        """timesteps [M] int64 -> [M, sum Cout]: sinusoidal embedding, time MLP (reference openaimodel.py:775-776) and the
        emb_layers projection of every ResBlock (266) as one batched GEMV.  A row depends on its own timestep only."""
        P = self._plan
        t_emb = ops.timestep_embedding(timesteps, self.model_channels, self.compute_dtype)
        e = ops.linear_small_m(t_emb, P["t0"].w, P["t0"].b, act_out=True)
        emb = ops.linear_small_m(e, P["t2"].w, P["t2"].b)
        return ops.linear_small_m(emb, P["emb_w"], P["emb_b"], act_in=True)

    def prepare_timesteps(self, steps):
        """Embedding rows of every timestep of one sampling, computed up front (16 timesteps per launch).  The rows are functions
        of the timestep alone -- not of x, not of the sample -- and a sampler knows its schedule before the first step, so the
        50 x 3 weight-streaming launches (56 MB each step) of the loop become 4 x 3 per sampling.  `forward` uses a row only when
        the sampler also names the step's timestep on the host (`_t_host`); any other caller gets the embedding computed from
        its `timesteps` tensor as before.  Row arithmetic does not depend on the number of rows in a launch: bit-identical."""
        self.prepare()
        if not EMB_TABLE:
            self._emb_table = {}
            return
        steps = [int(s_) for s_ in dict.fromkeys(int(s_) for s_ in steps)]
        dev = next(self.parameters()).device
        table = {}
        with torch.no_grad():
            for i in range(0, len(steps), 16):
                chunk = steps[i:i + 16]
                rows = self._embed(torch.tensor(chunk, device=dev, dtype=torch.int64))
                for j, s_ in enumerate(chunk):
                    table[s_] = rows[j]
        self._emb_table = table
        self._emb_table_plan = self._plan      # rows belong to THIS packing: a re-pack (changed weights) drops them (see _emb_rows)

    def _emb_rows(self, timesteps, N, out=None):
        """[N, sum Cout] embedding of this call: the precomputed row of the host-named timestep, else computed from `timesteps`."""
        # the hint is set by DDIMSampler.p_sample_ddim around ITS OWN apply_model calls only, with the host timestep its caller (the
        # sampling loop) named; any other caller of the UNet -- a corrector, a second model evaluation, another sampler sharing the
        # model -- runs with no hint and gets the embedding computed from its `timesteps` tensor
        hint = self.__dict__.get("_t_host")
        if self.__dict__.get("_emb_table_plan") is not self._plan:      # prepare() re-packed since the table was built: stale rows
            self._emb_table = {}
        row = self.__dict__.get("_emb_table", {}).get(hint) if hint is not None else None
        if row is not None and row.dtype == self.compute_dtype:
            rows = row.unsqueeze(0).expand(N, -1)
            return rows.contiguous() if out is None else out.copy_(rows)
        rows = self._embed(timesteps)
        return rows if out is None else out.copy_(rows)

    def _run_plan(self, x, timesteps, context, kv_cache=None, shared_prefix=False, c_input=None, emb_all=None):
        """x [N,Cin,H,W] fp32, timesteps [N] int64, context [N,L,D] fp16 -> eps [N,Cout,H,W] fp16.
        shared_prefix: x[:N/2] == x[N/2:] and timesteps likewise (see `cfg_shared_prefix`).
        c_input: optional [N, model_channels, H, W (or W/2: right half)] added to the output of the first input block
        (reference NVS_ldm.py:64-68).  emb_all: [N, sum Cout] embedding rows if the caller has them (`_emb_rows`)."""
        P = self._plan
        E = engine
        N, _, H, W = x.shape
        L = context.shape[1]
        ctx = context.reshape(N * L, context.shape[2])
        if emb_all is None:
            emb_all = self._embed(timesteps)       # [N, sum Cout]

        # Training: the reference's `use_checkpoint` (set by every LeftRefill config) keeps only each block's inputs and
        # recomputes the block in the backward (CheckpointFunction, ldm/modules/diffusionmodules/util.py:102-151) -- a memory
        # optimisation with identical results.  With 288 GB of HBM the activations simply stay resident (batch 16 at 256x512:
        # 9.2 GB peak, 30.2 ms per step); `recompute_in_backward = True` restores the reference's trade (7.6 GB, 39.4 ms).
        recompute = (self.recompute_in_backward and self.use_checkpoint and torch.is_grad_enabled()
                     and context.requires_grad)
        if kv_cache is None and TRAIN_KV_BATCH and torch.is_grad_enabled() and context.requires_grad and P["tblocks"]:
            # training: the [k;v] projections of the context for EVERY cross-attention as one GEMM (N = sum 2C = 24960 for the SD2 UNet)
            # and, in the backward, one input-gradient GEMM over the concatenated d[k;v] (K = 24960) instead of 16 + 16 launches on
            # 77 x batch rows each (24 us apiece at 16 x 77 rows) and 15 fan-in adds of the context gradient
            w_all = P.get("kv_all_w")
            if w_all is None or w_all.dtype != self.compute_dtype:
                w_all = P["kv_all_w"] = torch.cat([pt.attn2.kv.w for pt in P["tblocks"]], 0).contiguous()
            kv_all = ops.gemm_conv(ctx, w_all, B=1, H=1, W=N * L, taps=1)
            parts = ops.split_cols(kv_all, [pt.attn2.kv.w.shape[0] for pt in P["tblocks"]])
            kv_cache = [(p_, None) for p_ in parts]

        def ckpt(fn, act, *tensors):
            if not (recompute and (act.tok.requires_grad or any(t_.requires_grad for t_ in tensors))):
                return fn(act, *tensors)
            from torch.utils.checkpoint import checkpoint as torch_checkpoint
            n_, h_, w_ = act.N, act.H, act.W
            shape = {}

            gs_, gs2_ = act.gs, act.gs2      # producer statistics (no gradient flows through them): the recomputed block takes
                                             # exactly the path -- fused GroupNorm / GroupNorm-folded proj_in -- of the plain forward

            def body(tok, tok2, *ts):
                out = fn(E.Act(tok, n_, h_, w_, tok2=tok2, gs=gs_, gs2=gs2_), *ts)
                shape["hw"] = (out.H, out.W)
                # the block's own epilogue statistics leave the checkpoint with its output (round 6: the differentiable GroupNorm of the
                # next block takes them, exactly like the plain forward -- the two modes stay bit-identical)
                g_ = out.gs if out.tok2 is None else None
                shape["gs"] = None if g_ is None else (g_[1], g_[3])
                return (out.materialize(),) + ((None, None) if g_ is None else (g_[0], g_[2]))

            y, part, gp = torch_checkpoint(body, act.tok, act.tok2, *tensors, use_reentrant=False)
            gs_out = None if shape["gs"] is None else (part, shape["gs"][0], gp, shape["gs"][1])
            return E.Act(y, n_, *shape["hw"], gs=gs_out)

        def run(steps, act, half=False):
            """half: `act` is the first half of a CFG batch with identical halves; a SpatialTransformer ends that state."""
            for kind, p in steps:
                if kind in ("res", "conv") and half:      # planned like the full batch: identical partial sums
                    with E.plan_batch_scale(2):
                        act = (E.resblock(act, p, emb_all[:act.N, p.emb_off:p.emb_off + p.cout]) if kind == "res"
                               else E.conv(act, p, gn_stats=True))
                elif kind == "res":
                    emb_p = emb_all[:act.N, p.emb_off:p.emb_off + p.cout]      # (both halves share the timesteps)
                    act = ckpt(lambda a_, p=p, emb_p=emb_p: E.resblock(a_, p, emb_p), act)
                elif kind == "st":
                    act = ckpt(lambda a_, c_, p=p, d_=half: E.spatial_transformer(a_, c_, L, p, kv_cache, dup=d_), act, ctx)
                    half = False
                elif kind == "down":
                    act = E.conv(act, p, gn_stats=True)
                elif kind == "up":
                    act = E.conv(act, p, up=1, gn_stats=True)
                elif kind == "conv":
                    act = E.conv(act, p, gn_stats=True)
            return act

        taps = self.__dict__.get("_lr_taps")   # debugging hook: {name: NCHW fp32 block output} (eager mode only)

        def tap(name, a):
            if taps is not None:
                taps[name] = E.act_to_nchw(a, dtype=torch.float32)

        # CFG batch with identical halves: blocks 0 and 1 (up to the first cross-attention) run on the first half only
        shared = (shared_prefix and N % 2 == 0 and len(P["input"]) > 1 and len(P["input"][0]) == 1
                  and P["input"][0][0][0] == "conv" and [k for k, _ in P["input"][1]] == ["res", "st"]
                  and E.st_dup_ok(P["input"][1][1][1]) and not recompute)
        if shared:
            Nh = N // 2
            act = E.Act(ops.nchw_to_nhwc(x[:Nh], cpad=P["cin_pad"], dtype=self.compute_dtype), Nh, H, W)
        else:
            act = E.Act(ops.nchw_to_nhwc(x, cpad=P["cin_pad"], dtype=self.compute_dtype), N, H, W)
        hs = []
        for i, steps in enumerate(P["input"]):
            act, bstate = self._block_in(act, steps)
            act = run(steps, act, half=shared and i < 2)
            if i == 0 and c_input is not None:
                act = self._add_c_input(act, c_input)
            act = self._block_out(act, bstate)
            if shared and i == 0:      # the skip connection of the last output block wants the full batch
                hs.append(E.Act(E.dup2(act.tok), N, act.H, act.W,
                                gs=None if act.gs is None else (E.dup2(act.gs[0]), act.gs[1],
                                                                None if act.gs[2] is None else E.dup2(act.gs[2]), act.gs[3])))
            else:
                hs.append(act)
            tap(f"in{i}", hs[-1])
        act, bstate = self._block_in(act, P["middle"])
        act = self._block_out(run(P["middle"], act), bstate)
        tap("mid", act)
        for i, steps in enumerate(P["output"]):
            skip = hs.pop()
            act, bstate = self._block_in(E.Act(act.tok, act.N, act.H, act.W, tok2=skip.tok, gs=act.gs, gs2=skip.gs), steps)   # virtual th.cat([h, hs.pop()], 1)
            act = self._block_out(run(steps, act), bstate)
            tap(f"out{i}", act)
        pn, pc = P["out_norm"], P["out_conv"]
        if (act.tok2 is None and act.gs is not None and act.gs[2] is not None and E.gn_fuse_ok(act.tok) and pc.taps == 9
                and ops.gn_conv_out_ok(act.H, act.W, act.tok.shape[1], self.out_channels)):
            # `self.out` (reference 714-718, 812) in one launch: GroupNorm + SiLU + conv to 4 channels + NCHW, input read once
            return ops.gn_conv_out(act.tok, N, act.H, act.W, pn.g, pn.b, pn.eps, act.gs[2], act.gs[3], pc.w, pc.b, self.out_channels)
        act = E.gn(act, pn, True)
        act = E.conv(act, pc)
        return ops.nhwc_to_nchw(act.tok, N, act.H, act.W, self.out_channels)

    def _needs_autograd(self, context, c_input=None):
        return torch.is_grad_enabled() and (context.requires_grad or (c_input is not None and c_input.requires_grad))

    def _add_c_input(self, act, c_input):
        """h += c_input, or its right half when c_input is half as wide (NVS_ldm.py:64-68).  Layout conversion + one add of a
        [N, H, W, C] tensor per forward (c_input is constant over the DDIM loop); differentiable (torch ops on the token tensor)."""
        E = engine
        N, H, W = act.N, act.H, act.W
        tok = act.materialize().reshape(N, H, W, -1)
        c = c_input.to(tok.dtype).permute(0, 2, 3, 1)
        if tuple(c_input.shape[2:]) == (H, W):
            tok = tok + c
        else:
            tok = torch.cat([tok[:, :, :W // 2], tok[:, :, W // 2:] + c], dim=2)
        return E.Act(tok.reshape(N * H * W, -1).contiguous(), N, H, W)

    def forward(self, x, timesteps=None, context=None, y=None, **kwargs):
        """eps = UNet(x, t, context).  Returns fp16 (the reference's output dtype under autocast, appendix B)."""
        assert y is None, "must specify y if and only if the model is class-conditional"
        self.prepare()
        x = x.float().contiguous()
        timesteps = timesteps.to(torch.int64).contiguous()
        ctx_src = context
        if torch.is_grad_enabled() and x.requires_grad:
            raise NotImplementedError("gradient w.r.t. the noisy latent is not produced (p_losses feeds x_noisy without grad)")
        context = context.to(self.compute_dtype).contiguous()
        c_input = kwargs.get("c_input")
        if self._needs_autograd(context, c_input):
            # training (frozen weights, gradient flows to `context` and / or to the refinement branch behind `c_input`,
            # NVS_ldm.py:64-68): eager launches through leftrefill_amd.train_ops, torch.autograd records the HIP backward
            # kernels; no hipGraph, no K/V cache
            return self._run_plan(x, timesteps, context, c_input=c_input)
        if not self.use_hip_graph or c_input is not None or getattr(self, "eager_only", False):
            # eager inference: the same launches as the captured step (per-context K / V operands computed first), so the
            # two modes stay bit-identical.  (c_input / separator tokens of the NVS UNet: eager only.)
            with torch.no_grad():
                return self._run_plan(x, timesteps, context, self._context_kv(context), c_input=c_input,
                                      emb_all=self._emb_rows(timesteps, x.shape[0]))
        shared = bool(self.cfg_shared_prefix)
        key = (tuple(x.shape), tuple(context.shape), x.device.index, self.compute_dtype, shared, getattr(self, "_graph_slot", 0),
               bool(engine.MV_SHARDED))
        g = self._graphs.pop(key, None)
        if g is None:
            g = _StepGraph(self, x, timesteps, context, shared)
            while len(self._graphs) >= self.MAX_GRAPHS:      # LRU: dicts keep insertion order, re-inserted on every use
                self._graphs.pop(next(iter(self._graphs)))
        self._graphs[key] = g
        return g.replay(x, timesteps, context, ctx_src)


class _StepGraph:
    """One captured hipGraph of the UNet forward for a fixed (x, context) shape."""

    def __init__(self, model, x, t, ctx, shared_prefix=False):
        self.model = model
        self.x = x.clone()
        self.t = t.clone()
        self.ctx = ctx.clone()
        self.ctx_src = None       # (tensor object, _version) the K/V cache was computed from
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            self.kv = model._context_kv(self.ctx)                  # static per-context buffers
            self.emb = model._embed(self.t).contiguous()           # static buffer of the step's embedding rows, filled per replay
            model._run_plan(self.x, self.t, self.ctx, self.kv, shared_prefix, emb_all=self.emb)   # warm-up: kernel attributes
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.out = model._run_plan(self.x, self.t, self.ctx, self.kv, shared_prefix, emb_all=self.emb)

    def replay(self, x, t, ctx, ctx_src):
        self.x.copy_(x)
        # the embedding rows are inputs of the captured step: a sampler's precomputed row (UNetModel.prepare_timesteps), else
        # the same three launches the eager path runs, in front of the replay
        self.model._emb_rows(t, x.shape[0], out=self.emb)
        # Same context tensor object, not modified in place since last time -> K/V projections are still valid.
        # Holding a reference to the source tensor keeps its storage alive, so identity cannot be a recycled address.
        if self.ctx_src is None or self.ctx_src[0] is not ctx_src or self.ctx_src[1] != ctx_src._version:
            self.ctx.copy_(ctx)
            self.model._context_kv(self.ctx, out=self.kv)
            self.ctx_src = (ctx_src, ctx_src._version)
        self.graph.replay()
        return self.out.clone()
