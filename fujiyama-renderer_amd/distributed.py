"""Tile sharding across the GPUs of one node.

Tiles are independent units (the pixel-sample RNG restarts per tile and tiles
write disjoint framebuffer rectangles, reference src/fj_fixed_grid_sampler.cc:41-42,
src/fj_renderer.cc:976-995), so the frame is sharded with no data-path
collective: tile t belongs to rank t % G (row-major interleave), or -- `deal_tiles`
with the tile grid's width -- tile (tx, ty) to rank (tx + s ty) % G, a lattice whose
points lie evenly in BOTH directions whatever the width of the frame; the scene +
BLAS are replicated, and the only exchange is ONE gather of the finished RGBA
tiles to rank 0 (torch.distributed over RCCL/xGMI: equal-size slabs,
`dist.gather`).  Where frames repeat, `TileBalance` moves tiles from the slowest
ranks to the fastest between frames, from the ranks' own measured frame times.

Everything here works on CPU tensors with the gloo backend too (tests).
"""
import math

import torch
import torch.distributed as dist


def tiles_of_rank(n_tiles, rank, world):
    """Interleaved deal: tile ids owned by `rank`."""
    return list(range(rank, n_tiles, world))


def _lattice_gap(s, world):
    """squared length of the shortest vector between two tiles of one rank under the deal (tx + s ty) % world"""
    return min(x * x + y * y for y in range(0, world + 1) for x in range(-world, world + 1)
               if (x or y) and (x + s * y) % world == 0)


def lattice_step(world):
    """the row step s of the deal (tx + s ty) % world: the s whose lattice {(x, y): x + s y = 0 mod world} has the longest
    shortest vector -- the tiles of a rank then lie as far from each other as world tiles per rank allow"""
    return max(range(1, max(2, world)), key=lambda s: (_lattice_gap(s, world), -s))


def deal_tiles(n_tiles, world, nx=None, how="auto"):
    """the tile lists of every rank.  how = "interleave": tile t to rank t % world (what tiles_of_rank gives);
    "lattice" (needs nx, the tiles per row of the frame): tile (tx, ty) to rank (tx + s ty) % world, s = lattice_step(world).
    The interleave IS a lattice, with s = nx % world -- vertical stripes when the row length is a multiple of the rank count --
    and "auto" keeps it unless its tiles lie closer than 0.7 of what the best lattice gives (C3, 60 tiles per row on 8 ranks:
    s = 4 against 3, gaps 2 and 2.8 tiles: kept; the static lattice measured 3 % behind it on that frame, 5.89 against 6.04
    projected, and level with it once the feedback has run)."""
    if how not in ("auto", "interleave", "lattice"):
        raise ValueError("unknown tile deal %r" % (how,))
    if how == "auto" and nx and world > 1:
        s0 = nx % world
        poor = s0 == 0 or 2 * _lattice_gap(s0, world) < _lattice_gap(lattice_step(world), world)
        how = "lattice" if poor else "interleave"
    if how != "lattice" or not nx or world <= 1:
        return [tiles_of_rank(n_tiles, r, world) for r in range(world)]
    s = lattice_step(world)
    lists = [[] for _ in range(world)]
    for t in range(n_tiles):
        lists[(t % nx + s * (t // nx)) % world].append(t)
    return lists


def slab_capacity(n_tiles, world, slack=0.25):
    """tiles a rank's slab holds: the even share plus room for what TileBalance may hand to it"""
    even = int(math.ceil(n_tiles / float(world)))
    return even if world <= 1 else min(n_tiles, even + max(2, int(math.ceil(even * slack))))


def rebalance(lists, times, capacity, tile_share=0.8, threshold=1.5, noise=0.01, max_moves=None):
    """One round of the feedback deal: tiles move from the ranks with the longest measured frame times to those with the
    shortest, one at a time, while the predicted gap between the two is more than `threshold` tiles' worth and more than
    `noise` of the mean time (what two timings of the same work differ by).  A tile of rank r
    is priced at tile_share * times[r] / len(lists[r]) (the rest of a rank's time is per-launch cost that does not move with
    the tile).  Deterministic in its arguments: every rank runs it on the same all-gathered times and gets the same lists.
    Returns new lists (the input is not modified)."""
    world = len(lists)
    out = [list(l) for l in lists]
    if world <= 1:
        return out
    pred = [float(t) for t in times]
    price = [tile_share * pred[r] / max(1, len(out[r])) for r in range(world)]
    moved = [0] * world
    floor = noise * sum(pred) / world
    if max_moves is None:
        max_moves = max(1, sum(len(l) for l in out) // (2 * world))
    for _ in range(max_moves):
        order = sorted(range(world), key=lambda r: (pred[r], r))
        src = order[-1]
        dst = next((r for r in order if len(out[r]) < capacity), None)
        if dst is None or dst == src or len(out[src]) <= 1:
            break
        v = price[src]
        if pred[src] - pred[dst] <= max(threshold * v, floor):
            break
        # (which tile: spread over the donor's list, so that a rank does not lose a whole corner of the frame)
        at = (moved[src] * 7 + 3) % len(out[src])
        out[dst].append(out[src].pop(at))
        moved[src] += 1
        pred[src] -= v
        pred[dst] += v
    return [sorted(l) for l in out]


class TileBalance(object):
    """Feedback over repeated frames: `lists` starts as the static deal.  Each of the first `frames` frames ends with
    `update(times)` (the ranks' frame times, all-gathered by the caller); every `per_round` frames -- a rank's time for a deal
    is the SHORTEST of them: a frame in ten runs a millisecond or two long for reasons that are not its tiles -- the tiles are
    re-dealt with `rebalance`; the frame after the last adaptation frame runs, and every later one keeps, the deal that had
    the shortest slowest-rank time."""

    def __init__(self, lists, capacity, frames=8, per_round=2):
        self.lists = [list(l) for l in lists]
        self.capacity = capacity
        self.per_round = max(1, per_round)
        self.rounds_left = max(0, frames) // self.per_round if len(lists) > 1 else 0
        self.seen = []                         # times of the current deal, one list per frame
        self.best = None                       # (slowest rank's time, lists)
        self.history = []                      # slowest rank's time of every deal tried

    @property
    def adapting(self):
        return self.rounds_left > 0

    def update(self, times):
        if self.rounds_left <= 0:
            return self.lists
        self.seen.append([float(t) for t in times])
        if len(self.seen) < self.per_round:
            return self.lists
        times = [min(f[r] for f in self.seen) for r in range(len(self.lists))]
        self.seen = []
        worst = max(times)
        self.history.append(worst)
        if self.best is None or worst < self.best[0]:
            self.best = (worst, [list(l) for l in self.lists])
        self.rounds_left -= 1
        self.lists = rebalance(self.lists, times, self.capacity) if self.rounds_left > 0 else [list(l) for l in self.best[1]]
        return self.lists


def _grid(xres, yres, tile_w, tile_h):
    return int(math.ceil(xres / float(tile_w))), int(math.ceil(yres / float(tile_h)))


def pack_tiles(fb, tile_ids, tile_w, tile_h):
    """fb [H, W, 4] -> slab [len(tile_ids), tile_h, tile_w, 4] (edge tiles zero padded).

    Assumes the render region is the full frame (tile id = row-major grid index).
    """
    H, W, C = fb.shape
    nx, ny = _grid(W, H, tile_w, tile_h)
    padded = torch.zeros((ny * tile_h, nx * tile_w, C), dtype=fb.dtype, device=fb.device)
    padded[:H, :W] = fb
    tiles = padded.view(ny, tile_h, nx, tile_w, C).permute(0, 2, 1, 3, 4).reshape(ny * nx, tile_h, tile_w, C)
    idx = torch.as_tensor(tile_ids, dtype=torch.long, device=fb.device)
    return tiles.index_select(0, idx).contiguous()


def unpack_tiles(slabs, tile_id_lists, xres, yres, tile_w, tile_h):
    """Inverse of pack_tiles for the slabs of all ranks -> fb [H, W, 4]."""
    C = slabs[0].shape[-1]
    nx, ny = _grid(xres, yres, tile_w, tile_h)
    tiles = torch.zeros((ny * nx, tile_h, tile_w, C), dtype=slabs[0].dtype, device=slabs[0].device)
    for slab, ids in zip(slabs, tile_id_lists):
        n = len(ids)
        if n:
            idx = torch.as_tensor(ids, dtype=torch.long, device=slab.device)
            tiles.index_copy_(0, idx, slab[:n])
    padded = tiles.view(ny, nx, tile_h, tile_w, C).permute(0, 2, 1, 3, 4).reshape(ny * tile_h, nx * tile_w, C)
    return padded[:yres, :xres].contiguous()


def _backend_is_device_capable():
    """RCCL ("nccl") moves device tensors; gloo gets host copies of the slabs (CPU tests, bench.py --dry-ranks)"""
    try:
        return dist.get_backend() == "nccl"
    except Exception:  # noqa: BLE001
        return False


class _DeviceSlabs(object):
    """per-process cache of the device-side plumbing of gather_frame: the tile rectangles of this rank and -- on rank 0 --
    of every rank in gather order (padding entries have zero area), the send slab and the receive buffer.  Packing and
    scattering are ONE launch each of the core's k_move_tiles (fjgpu_pack_tiles / fjgpu_unpack_tiles); nothing is
    allocated per frame."""

    def __init__(self, fb, rects, n_tiles, tile_w, tile_h, rank, world, lists=None, capacity=None):
        dev = fb.device
        if lists is None:
            lists = [tiles_of_rank(n_tiles, r, world) for r in range(world)]
        self.per_rank = int(capacity) if capacity else max(len(l) for l in lists)
        self.tile_px = tile_w * tile_h

        def table(ids):
            rows = [list(rects[t]) for t in ids] + [[0, 0, 0, 0]] * (self.per_rank - len(ids))
            return torch.tensor(rows, dtype=torch.int32, device=dev).contiguous()
        self.mine = table(lists[rank])
        self.slab = torch.zeros((self.per_rank, tile_h, tile_w, 4), dtype=torch.float32, device=dev)
        self.all = self.recv = None
        if rank == 0:
            self.all = torch.cat([table(lists[r]) for r in range(world)], dim=0).contiguous()
            self.recv = torch.zeros((world, self.per_rank, tile_h, tile_w, 4), dtype=torch.float32, device=dev)

    def retable(self, rects, rank, lists):
        """another deal of the same tiles over the same slabs (TileBalance): only the rectangle tables change"""
        dev = self.slab.device

        def table(ids):
            assert len(ids) <= self.per_rank
            rows = [list(rects[t]) for t in ids] + [[0, 0, 0, 0]] * (self.per_rank - len(ids))
            return torch.tensor(rows, dtype=torch.int32, device=dev).contiguous()
        self.mine = table(lists[rank])
        if self.all is not None:
            self.all = torch.cat([table(l) for l in lists], dim=0).contiguous()


def gather_frame(fb, n_tiles, tile_w, tile_h, rank, world, mode=None, rects=None, lists=None, capacity=None):
    """Gather every rank's finished tiles to rank 0; returns the assembled
    framebuffer on rank 0 and None elsewhere.  One exchange per frame:
    33.2 MB total at 1080p, <= 4.1 MB per peer.

    mode (default: env FJ_GATHER or "gather"):
      "gather"      one dist.gather of equal-size slabs to rank 0 (RCCL: grouped send / recv over xGMI)
      "send_recv"   the same exchange spelled as point-to-point dist.send / dist.recv
      "all_gather"  dist.all_gather of the slabs (every rank receives the frame; rank 0 uses it)
    If the backend refuses "gather" before any data moves (RuntimeError: not supported), the
    point-to-point form is used for this and every later frame.

    lists / capacity: the deal (tile ids per rank, `deal_tiles` / `TileBalance.lists`; default: the interleave) and the
    tiles a slab holds (default: the longest list) -- the same on every rank.  `rects` and `lists` are immutable to the
    caller: a new deal is a new object (the device-side tables are rebuilt when the object or its fingerprint changes)."""
    import os
    H, W, _ = fb.shape
    if world == 1:
        return fb
    mode = mode or _state.get("mode") or os.environ.get("FJ_GATHER", "gather")
    if lists is None:
        lists = [tiles_of_rank(n_tiles, r, world) for r in range(world)]
    per_rank = int(capacity) if capacity else max(len(l) for l in lists)
    mine = lists[rank]
    dev_path = fb.is_cuda and rects is not None          # (device framebuffer: the core's own pack / scatter kernels)
    ds = None
    if dev_path:
        from . import gpu
        # (the rectangle table and the frame size are part of the key: another region or resolution with the same tile count
        # and the same framebuffer allocation must not scatter with the old tables)
        key = (fb.data_ptr(), H, W, n_tiles, tile_w, tile_h, rank, world, per_rank)
        # (per frame this is two identity tests: the tables are compared by VALUE only when the caller hands over another
        # object -- hashing 2040 rectangles and every tile id cost ~1 ms of every timed frame on every rank)
        # `rects` and `lists` are to be treated as IMMUTABLE by the caller (TileBalance builds new lists for every deal).  A caller that edits
        # the same object in place is still caught by a fingerprint that costs O(world) per frame: lengths, and first / last entry of every list.
        rfp = (len(rects), tuple(rects[0]) if len(rects) else None, tuple(rects[-1]) if len(rects) else None)
        lfp = tuple((len(l), l[0] if l else -1, l[-1] if l else -1, l[len(l) // 2] if l else -1) for l in lists)
        if _state.get("rects_ref") is not rects or _state.get("rects_fp") != rfp:
            rv = tuple(tuple(int(v) for v in r) for r in rects)
            if _state.get("rects_val") != rv:
                _state["slabs_key"] = None
            _state["rects_ref"], _state["rects_val"], _state["rects_fp"] = rects, rv, rfp
        deal_changed = False
        if _state.get("lists_ref") is not lists or _state.get("lists_fp") != lfp:
            lv = tuple(tuple(l) for l in lists)
            deal_changed = _state.get("lists_val") != lv
            _state["lists_ref"], _state["lists_val"], _state["lists_fp"] = lists, lv, lfp
        if _state.get("slabs_key") != key:
            _state["slabs"], _state["slabs_key"] = _DeviceSlabs(fb, rects, n_tiles, tile_w, tile_h, rank, world, lists, per_rank), key
            deal_changed = False
        ds = _state["slabs"]
        if deal_changed:
            ds.retable(rects, rank, lists)
        stream = torch.cuda.current_stream(fb.device).cuda_stream
        gpu.pack_tiles(fb.data_ptr(), W, ds.mine.data_ptr(), per_rank, ds.tile_px, ds.slab.data_ptr(), stream)
        slab = ds.slab
    else:
        slab = torch.zeros((per_rank, tile_h, tile_w, fb.shape[-1]), dtype=fb.dtype, device=fb.device)
        if mine:
            slab[:len(mine)] = pack_tiles(fb, mine, tile_w, tile_h)
    on_device = _backend_is_device_capable()
    if not on_device:
        slab = slab.cpu()
    out = None
    recv = list(ds.recv.unbind(0)) if (ds is not None and rank == 0 and on_device) else None      # views of ONE buffer
    if mode == "gather":
        try:
            if rank == 0:
                out = recv if recv is not None else [torch.empty_like(slab) for _ in range(world)]
                dist.gather(slab, gather_list=out, dst=0)
            else:
                dist.gather(slab, gather_list=None, dst=0)
        except (RuntimeError, NotImplementedError) as e:      # raised at dispatch, on every rank alike
            if "support" not in str(e).lower() and "implement" not in str(e).lower():
                raise
            mode = _state["mode"] = "send_recv"
            out = None
    if mode == "send_recv":
        if rank == 0:
            if recv is not None:
                recv[0].copy_(slab)
                out = recv
            else:
                out = [slab] + [torch.empty_like(slab) for _ in range(world - 1)]
            for r in range(1, world):
                dist.recv(out[r], src=r)
        else:
            dist.send(slab, dst=0)
    elif mode == "all_gather":
        got = recv if recv is not None else [torch.empty_like(slab) for _ in range(world)]
        dist.all_gather(got, slab)
        out = got if rank == 0 else None
    elif mode != "gather":
        raise ValueError("unknown gather mode %r" % (mode,))
    if rank != 0:
        return None
    if ds is not None:
        # scatter into rank 0's own framebuffer (its own tiles are already there): one launch over all ranks' tiles
        from . import gpu
        if recv is None or out is not recv:              # (host-side exchange, gloo: bring the slabs back to the device)
            for r in range(world):
                ds.recv[r].copy_(out[r])
        gpu.unpack_tiles(fb.data_ptr(), W, ds.all.data_ptr(), world * per_rank, ds.tile_px, ds.recv.data_ptr(),
                         torch.cuda.current_stream(fb.device).cuda_stream)
        return fb
    frame = unpack_tiles(out, lists, W, H, tile_w, tile_h)
    return frame.to(fb.device) if frame.device != fb.device else frame


def share_times(ms, rank, world, device=None):
    """every rank's frame time on every rank (the input of TileBalance.update): one all_gather of a double"""
    if world <= 1:
        return [float(ms)]
    dev = device if (device is not None and _backend_is_device_capable()) else torch.device("cpu")
    mine = torch.tensor([float(ms)], dtype=torch.float64, device=dev)
    got = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(got, mine)
    return [float(g.item()) for g in got]


_state = {}
