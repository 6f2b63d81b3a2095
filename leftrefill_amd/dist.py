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
# Reference semantics (ldm/modules/multiview_attention.py:440-460, concat_target=True): the self-attention sequence of a
# sample is [target(canvas 0), ref_0, ..., ref_{v-1}] ((v+1) s^2 tokens); afterwards the target rows are written to the
# right half of EVERY canvas and ref_i to canvas i's left half.  Sharded form: every rank all-gathers the raw canvases
# (one collective per transformer block), forms K/V for the whole sequence, computes Q / attention / out-projection only
# for its own rows [target, ref_rank] (the target rows are replicated work, bit-identical on every rank, so no second
# exchange is needed for the write-back), and rebuilds its own canvas.  Everything else in the UNet is canvas-local.
# The helpers below are device-agnostic (torch + torch.distributed only) so the index logic is covered by gloo tests.
# ---------------------------------------------------------------------------------------------------------------
def mv_group_size():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def mv_all_gather_canvases(x_local):
    """x_local [b, T, C] (this rank's canvas of each of the b local sample groups) -> [b, v, T, C] in rank order."""
    world = mv_group_size()
    if world == 1:
        return x_local[:, None]
    if x_local.is_cuda and dist.get_backend() == "gloo":
        # test transport only (ranks sharing one GPU, tests/test_gpu_unet.py): gloo has no device all_gather -> stage on host
        host = x_local.detach().cpu().contiguous()
        bufs = [torch.empty_like(host) for _ in range(world)]
        dist.all_gather(bufs, host)
        return torch.stack(bufs, dim=1).to(x_local.device)
    bufs = [torch.empty_like(x_local) for _ in range(world)]
    dist.all_gather(bufs, x_local.contiguous())
    return torch.stack(bufs, dim=1)


def mv_sequence_from_canvases(x_all, s):
    """[b, v, s*2s, C] canvases -> [b, (v+1)*s*s, C] sequence [target(canvas 0), ref_0 .. ref_{v-1}] (pure torch)."""
    b, v, T, C = x_all.shape
    g = x_all.reshape(b, v, s, 2 * s, C)
    seq = torch.cat([g[:, 0:1, :, s:, :], g[:, :, :, :s, :]], dim=1)
    return seq.reshape(b, (v + 1) * s * s, C)


def mv_own_rows(seq, rank, s):
    """Rows this rank owns as queries: the target block and its own reference block -> [b, 2*s*s, C]."""
    s2 = s * s
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
