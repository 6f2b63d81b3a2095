"""Sample-sharded data parallelism for the sampler.

Samples are independent for all DDIM steps, so the batch dimension shards across ranks with NO collective inside the
loop: every rank runs its slice through the same replicated UNet (1.73 GB fp16 weights) and the final latents are
all-gathered once ([B_local, 4, h, w] fp32 = 131 KB per sample).  One process per GPU; `nccl` (= RCCL over xGMI) on
the GPU box, `gloo` in CPU tests.
"""
import torch
import torch.distributed as dist


def shard_range(total, rank, world):
    """Contiguous, balanced [start, stop) of `total` samples for `rank` (first `total % world` ranks get one more)."""
    base, rem = divmod(total, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def shard_tree(obj, rank, world, total=None):
    """Slice dim 0 of every tensor in a (possibly nested) dict / list conditioning structure."""
    if torch.is_tensor(obj):
        s, e = shard_range(obj.shape[0] if total is None else total, rank, world)
        return obj[s:e]
    if isinstance(obj, dict):
        return {k: shard_tree(v, rank, world, total) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return type(obj)(shard_tree(v, rank, world, total) for v in obj)
    return obj


def all_gather_cat(x, total=None):
    """Concatenate per-rank results along dim 0 in rank order (ragged shards allowed)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return x
    world = dist.get_world_size()
    total = total if total is not None else None
    sizes = [torch.zeros(1, dtype=torch.long, device=x.device) for _ in range(world)]
    dist.all_gather(sizes, torch.tensor([x.shape[0]], dtype=torch.long, device=x.device))
    sizes = [int(s.item()) for s in sizes]
    mx = max(sizes)
    pad = torch.zeros((mx,) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
    pad[: x.shape[0]] = x
    outs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(outs, pad)
    return torch.cat([o[:n] for o, n in zip(outs, sizes)], dim=0)


def sample_sharded(sample_fn, cond, uncond, x_T, batch_size):
    """Run `sample_fn(cond_shard, uncond_shard, x_T_shard, local_batch)` on this rank's slice and gather all latents."""
    if not (dist.is_available() and dist.is_initialized()):
        return sample_fn(cond, uncond, x_T, batch_size)
    rank, world = dist.get_rank(), dist.get_world_size()
    s, e = shard_range(batch_size, rank, world)
    out = sample_fn(shard_tree(cond, rank, world, batch_size), shard_tree(uncond, rank, world, batch_size),
                    None if x_T is None else x_T[s:e], e - s)
    return all_gather_cat(out)


# ---------------------------------------------------------------------------------------------------------------
# Multi-view inference, one stitched canvas [ref_i | target] per rank.
#
# Reference semantics (ldm/modules/multiview_attention.py:436-462, concat_target=True): the self-attention sequence of a
# sample is [target (right half of canvas 0), ref_0, ..., ref_{v-1}] ((v+1) s^2 tokens); afterwards the target rows are
# written to the right half of EVERY canvas and ref_i to canvas i's left half.  Sharded form, per transformer block:
#   * exchange ONLY what the sequence is made of: an all-gather of the reference halves (s^2 tokens per rank, into one
#     preallocated buffer) and a broadcast of rank 0's target half -- the other ranks' target halves are never read;
#   * every rank builds K/V for the whole sequence, but Q / attention / out-projection only for its own rows
#     [target, ref_rank]; the target rows are replicated work with bit-identical inputs and a shape-static kernel plan
#     (leftrefill_amd.ops tile table), hence bit-identical results on every rank: no second exchange for the write-back.
# Round 5: ONE exchange of the inputs per block (mv_exchange_canvases: all-gather of whole canvases, LayerNorm statistics packed into
# the same message) and, by default, the TARGET query rows split over the ranks with one all-gather of the new target rows behind the
# out-projection (mv_own_rows_split / mv_gather_target): two collectives per block instead of four, and 1.25 s^2 instead of 2 s^2 query
# rows per rank at four views.  LEFTREFILL_MV_SPLIT_TARGET=0 keeps the replicated target rows (one collective per block).
# Everything else in the UNet is canvas-local.  The helpers are device-agnostic (torch + torch.distributed only) so the
# index logic is covered by gloo tests on CPU; on GPUs the backend is `nccl` (RCCL over xGMI).
# ---------------------------------------------------------------------------------------------------------------
# Per-rank cost of the sharded multi-view step without the other ranks (bench.py --workload mv5 --mv-shard --gpus 1): with
# LEFTREFILL_MV_SIM_WORLD = v and no process group, this process plays rank LEFTREFILL_MV_SIM_RANK (default 0) of a v-rank job --
# the sequence buffer is filled with local copies where the all-gather / broadcast would deliver the other ranks' rows (same
# shapes, same kernels, same bytes written; only the wire is missing).
def _sim_world():
    import os
    if dist.is_available() and dist.is_initialized():
        return 0
    return int(os.environ.get("LEFTREFILL_MV_SIM_WORLD", "0"))


def mv_group_size():
    if dist.is_available() and dist.is_initialized():
        return dist.get_world_size()
    return _sim_world() or 1


def mv_rank():
    import os
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank()
    return int(os.environ.get("LEFTREFILL_MV_SIM_RANK", "0")) if _sim_world() else 0


def _staged(x):
    """gloo has no device collectives: ranks that share one GPU in the tests stage through the host."""
    return x.is_cuda and dist.get_backend() == "gloo"


def mv_gather_sequence(x_local, s, seq=None):
    """x_local [b, 2*s*s, C]: this rank's canvas tokens (row-major s x 2s: left = ref_rank, right = its copy of the target)
    of each of the b local samples -> the re-arranged sequence [b, (v+1)*s*s, C] = [target of rank 0, ref_0 .. ref_{v-1}].
    `seq`: optional preallocated output (static buffer for hipGraph capture).  Two collectives, (v+1)*s*s*C elements
    received per sample instead of the 2*v*s*s*C of an all-gather of whole canvases."""
    b, T, C = x_local.shape
    s2 = s * s
    assert T == 2 * s2
    world = mv_group_size()
    rank = mv_rank()
    g = x_local.reshape(b, s, 2 * s, C)
    if seq is None:
        seq = torch.empty(b, world + 1, s2, C, dtype=x_local.dtype, device=x_local.device)
    else:
        seq = seq.reshape(b, world + 1, s2, C)
    ref = g[:, :, :s, :].reshape(b, s2, C).contiguous()
    tgt = g[:, :, s:, :].reshape(b, s2, C).contiguous()
    if world == 1:
        seq[:, 0] = tgt
        seq[:, 1] = ref
        return seq.reshape(b, 2 * s2, C)
    if _sim_world():          # simulated peers: what the collectives would write, written from local data
        seq[:, 0] = tgt
        seq[:, 1:] = ref[:, None]
        return seq.reshape(b, (world + 1) * s2, C)
    if _staged(x_local):
        ref_h, tgt_h = ref.cpu(), tgt.cpu()
        allr = torch.empty(world * b, s2, C, dtype=ref_h.dtype)           # concatenation form (the one gloo accepts)
        dist.all_gather_into_tensor(allr, ref_h)
        dist.broadcast(tgt_h, src=0)
        seq[:, 0] = tgt_h.to(x_local.device)
        seq[:, 1:] = allr.reshape(world, b, s2, C).permute(1, 0, 2, 3).to(x_local.device)
        return seq.reshape(b, (world + 1) * s2, C)
    if b == 1:          # the gathered blocks are already in sequence order: receive straight into the sequence buffer
        dist.all_gather_into_tensor(seq[0, 1:], ref)        # [world, s2, C] <- world x [1, s2, C]
        if rank == 0:
            seq[0, 0].copy_(tgt[0])
        dist.broadcast(seq[0, 0], src=0)
    else:
        allr = torch.empty(world * b, s2, C, dtype=ref.dtype, device=ref.device)
        dist.all_gather_into_tensor(allr, ref)
        dist.broadcast(tgt, src=0)
        seq[:, 0] = tgt
        seq[:, 1:] = allr.reshape(world, b, s2, C).permute(1, 0, 2, 3)
    return seq.reshape(b, (world + 1) * s2, C)


def mv_all_gather_rows(send):
    """send [R, ...] (contiguous) -> [world * R, ...] in rank order: ONE all_gather_into_tensor (RCCL on GPUs; gloo stages through the
    host; simulated peers receive copies of the local rows -- same bytes written, no wire)."""
    world = mv_group_size()
    if world == 1:
        return send
    if _sim_world():
        return send[None].expand((world,) + tuple(send.shape)).reshape((world * send.shape[0],) + tuple(send.shape[1:])).contiguous()
    src = send.contiguous()
    staged = _staged(src)
    if staged:
        src = src.cpu()
    out = torch.empty((world * src.shape[0],) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device)      # concatenation form
    dist.all_gather_into_tensor(out, src)
    return out.to(send.device) if staged else out


_MV_PLANS = {}


def mv_shard_plan(b, v, s, rank, split, device):
    """Index tables (int32, on `device`, cached per shape) of the sharded block's row copies (leftrefill_amd.ops.row_copy):
      seq_src [b Ls]  : received row (canvas-major message [v][b][2 s^2]) of every sequence row [target of canvas 0, ref_0 .. ref_{v-1}]
      own_src [b Lo]  : received row of every row this rank owns as a query ([target slice `rank`, ref_rank] with `split`, else
                        [target, ref_rank])
      and, with `split`, the write-back of the canvas [ref' | target'] from y [b Lo] (this rank's rows) and the all-gathered target
      slices [v][b][s^2 / v]:  ref_src / ref_dst,  tgt_src / tgt_dst  (b s^2 rows each)."""
    key = (b, v, s, rank, bool(split), str(device))
    p = _MV_PLANS.get(key)
    if p is not None:
        return p
    s2, T = s * s, 2 * s * s
    Ls = (v + 1) * s2
    ar = torch.arange
    row, col = ar(s2) // s, ar(s2) % s
    tgt_rows = row * (2 * s) + s + col          # canvas row of target pixel p
    ref_rows = row * (2 * s) + col              # canvas row of reference pixel p

    def recv(canvas, bi, crow):
        return (canvas * b + bi) * T + crow

    seq_src = torch.empty(b, v + 1, s2, dtype=torch.long)
    for bi in range(b):
        seq_src[bi, 0] = recv(0, bi, tgt_rows)
        for j in range(v):
            seq_src[bi, 1 + j] = recv(j, bi, ref_rows)
    n = s2 // v if split else s2
    own = torch.cat([seq_src[:, 0, rank * n:(rank + 1) * n] if split else seq_src[:, 0], seq_src[:, 1 + rank]], dim=1)
    p = {"seq_src": seq_src.reshape(-1), "own_src": own.reshape(-1), "Lo": n + s2, "n": n}
    if split:
        Lo = n + s2
        bi = ar(b)[:, None]
        p["ref_src"] = (bi * Lo + n + ar(s2)[None]).reshape(-1)
        p["ref_dst"] = (bi * T + ref_rows[None]).reshape(-1)
        pp = ar(s2)[None]
        p["tgt_src"] = (((pp // n) * b + bi) * n + pp % n).reshape(-1)
        p["tgt_dst"] = (bi * T + tgt_rows[None]).reshape(-1)
    p = {k: (t.to(device=device, dtype=torch.int32).contiguous() if torch.is_tensor(t) else t) for k, t in p.items()}
    _MV_PLANS[key] = p
    return p


def mv_exchange_canvases(x_local, extra=None):
    """ONE collective per transformer block (round 5, VERDICT r4 #5a): all-gather of the ranks' WHOLE canvases, with an optional
    per-row fp32 payload (the rows' LayerNorm partial sums) packed behind each row's bytes so that it rides in the same message.
    x_local [b, T, C] 16-bit, extra [b, T, E] fp32 or None  ->  (x_all [b, v, T, C], extra_all [b, v, T, E] | None), views into the
    receive buffer (row pitch 2 C + 4 E bytes; consumers copy what they need into sequence order, mv_sequence_from_canvases).
    Against mv_gather_sequence (all-gather of the reference halves + broadcast of rank 0's target half, twice when the statistics
    travel too: four collectives) this sends the ranks' unused target halves as well -- 2 v s^2 rows instead of (v + 1) s^2 -- and
    needs one launch of the collective instead of four."""
    b, T, C = x_local.shape
    world = mv_group_size()
    if world == 1:
        return x_local[:, None], (None if extra is None else extra[:, None])
    if _sim_world():          # simulated peers: what the collective would deliver, written from local data (same bytes, no wire)
        return (x_local[:, None].expand(b, world, T, C).contiguous(),
                None if extra is None else extra[:, None].expand(b, world, T, extra.shape[2]).contiguous())
    E = 0 if extra is None else extra.shape[2]
    esz = x_local.element_size()
    rb = C * esz + 4 * E
    assert esz == 2 and (C * esz) % 4 == 0
    send = torch.empty(b * T, rb, dtype=torch.uint8, device=x_local.device)
    send[:, :C * esz].view(x_local.dtype).copy_(x_local.reshape(b * T, C))
    if E:
        assert extra.dtype == torch.float32 and extra.shape[:2] == (b, T)
        send[:, C * esz:].view(torch.float32).copy_(extra.reshape(b * T, E))
    recv = mv_all_gather_rows(send).reshape(world, b * T, rb)
    x_all = recv[:, :, :C * esz].view(x_local.dtype).reshape(world, b, T, C).transpose(0, 1)
    e_all = recv[:, :, C * esz:].view(torch.float32).reshape(world, b, T, E).transpose(0, 1) if E else None
    return x_all, e_all


def mv_own_rows_split(seq, rank, s, world):
    """Rows this rank owns as queries when the TARGET rows are split over the ranks (VERDICT r4 #5b): slice `rank` of the target
    block (s^2 / world rows) and its own reference block -> [b, s^2 / world + s^2, C]."""
    s2 = s * s
    assert s2 % world == 0
    n = s2 // world
    return torch.cat([seq[:, rank * n:(rank + 1) * n], seq[:, (1 + rank) * s2:(2 + rank) * s2]], dim=1)


def mv_gather_target(y_t):
    """y_t [b, s^2 / world, C]: this rank's slice of the new target rows -> the whole target block [b, s^2, C] on every rank
    (one all-gather per block; every rank receives the same bytes, so the replicated right halves stay bit-identical)."""
    world = mv_group_size()
    if world == 1:
        return y_t
    b, n, C = y_t.shape
    out = mv_all_gather_rows(y_t.contiguous())
    return out.reshape(world, b, n, C).permute(1, 0, 2, 3).reshape(b, world * n, C)


def mv_all_gather_canvases(x_local):
    """x_local [b, T, C] -> [b, v, T, C] in rank order (whole canvases; kept for tests of the index helpers)."""
    world = mv_group_size()
    if world == 1:
        return x_local[:, None]
    src = x_local.detach().cpu().contiguous() if _staged(x_local) else x_local.contiguous()
    out = torch.empty((world * src.shape[0],) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device)
    dist.all_gather_into_tensor(out, src)
    return out.reshape((world,) + tuple(src.shape)).transpose(0, 1).contiguous().to(x_local.device)


def mv_sequence_from_canvases(x_all, s):
    """[b, v, s*2s, C] canvases -> [b, (v+1)*s*s, C] sequence [target(canvas 0), ref_0 .. ref_{v-1}] (pure torch)."""
    b, v, T, C = x_all.shape
    g = x_all.reshape(b, v, s, 2 * s, C)
    seq = torch.cat([g[:, 0:1, :, s:, :], g[:, :, :, :s, :]], dim=1)
    return seq.reshape(b, (v + 1) * s * s, C)


def mv_own_rows(seq, rank, s):
    """Rows this rank owns as queries: the target block and its own reference block -> [b, 2*s*s, C].  Rank 0's two blocks are
    adjacent in the sequence: a view (no copy; contiguous when b == 1)."""
    s2 = s * s
    if rank == 0:
        return seq[:, :2 * s2]
    return torch.cat([seq[:, :s2], seq[:, (1 + rank) * s2:(2 + rank) * s2]], dim=1)


def mv_canvas_from_own(y, s):
    """[b, 2*s*s, C] rows [target', ref'] -> canvas tokens [b, s*2s, C] (left = ref', right = target')."""
    b, _, C = y.shape
    t, r = y[:, :s * s].reshape(b, s, s, C), y[:, s * s:].reshape(b, s, s, C)
    return torch.cat([r, t], dim=2).reshape(b, 2 * s * s, C)


def allreduce_mean_grads(params):
    """Data-parallel training with a frozen backbone: only the prompt-token parameters carry gradients (the reference's DDP
    reduces every UNet / CLIP gradient although its optimizer owns just `special_embeddings`).  One flat
    all-reduce (RCCL over xGMI on GPUs) of the few trainable gradients, averaged like DDP."""
    grads = [p.grad for p in params if p.grad is not None]
    if not grads or not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return
    flat = torch.cat([g.reshape(-1).float() for g in grads])
    dist.all_reduce(flat)
    flat /= dist.get_world_size()
    off = 0
    for g in grads:
        n = g.numel()
        g.copy_(flat[off:off + n].reshape(g.shape).to(g.dtype))
        off += n


# ---------------------------------------------------------------------------------------------------------------
# Classifier-free guidance split over two ranks ("cond + uncond x2 batch"): when the user batch is
# smaller than the number of GPUs, rank 2j runs the UNCONDITIONAL pass and rank 2j + 1 the CONDITIONAL pass of the same
# samples (reference ddim.py:317-343 runs them as one batch of 2B); per DDIM step the two eps halves are exchanged with ONE
# all-gather inside the pair ([2, B, 4, h, w] fp16: 262 KB at B = 4, 64 x 128) and both ranks apply the same guided update
# (same noise: the pair shares its RNG seed).  World sizes > 2 form independent pairs (2 j, 2 j + 1).
# ---------------------------------------------------------------------------------------------------------------
_SPLIT_CFG = {"on": False, "group": None}


def enable_split_cfg(on=True):
    """Switch the sampler's split mode (collective: every rank must call it).  With no process group (world 1) the mode still
    runs -- the two passes go through the UNet one after the other at batch B -- which is what the single-GPU test covers."""
    _SPLIT_CFG["on"] = bool(on)
    _SPLIT_CFG["group"] = None
    if on and dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        world, rank = dist.get_world_size(), dist.get_rank()
        assert world % 2 == 0, "split CFG pairs ranks (2 j, 2 j + 1): the world size must be even"
        for j in range(world // 2):
            g = dist.new_group([2 * j, 2 * j + 1])
            if rank // 2 == j:
                _SPLIT_CFG["group"] = g


def split_cfg_active():
    return _SPLIT_CFG["on"]


def split_cfg_role():
    """0: this rank runs the unconditional pass, 1: the conditional pass, None: both (no process group)."""
    if _SPLIT_CFG["group"] is None:
        return None
    return dist.get_rank() % 2


def cfg_exchange(e_local):
    """e_local [B, ...] = this rank's eps half -> [2B, ...], unconditional half first, on both ranks of the pair."""
    g = _SPLIT_CFG["group"]
    assert g is not None
    e_local = e_local.contiguous()
    if _staged(e_local):
        h = e_local.cpu()
        out = torch.empty((2 * h.shape[0],) + tuple(h.shape[1:]), dtype=h.dtype)
        dist.all_gather_into_tensor(out, h, group=g)
        return out.to(e_local.device)
    out = torch.empty((2 * e_local.shape[0],) + tuple(e_local.shape[1:]), dtype=e_local.dtype, device=e_local.device)
    dist.all_gather_into_tensor(out, e_local, group=g)
    return out
