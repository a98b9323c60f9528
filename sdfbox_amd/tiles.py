"""Screen-tile sharding of one frame over the GPUs of a node (new design, the
reference is single-GPU: SURVEY.md 8e).

Pixels are independent and the scene is read-only, so the octree is replicated
on every device and the *frame* is sharded: row bands of `band_rows` rows are
dealt round-robin (band b -> rank b mod world), which balances sky against
object rows.  Each rank renders its bands into a compact local buffer of
`rows_per_rank` rows; one gather per frame brings the buffers to rank 0 over
xGMI (RCCL send/recv: every peer has its own direct link into rank 0), where
a small kernel restores row order.  No other collective is on the data path.

The gather is bound by one xGMI link per peer (frame bytes / world over ~77 GB/s
per direction), and at ~8 Gray/s per GPU an RGBA32F frame would make every
world size link-bound.  Every pixel the shader writes is (a, a, a, steps) or the
sky constant, so ranks render 5-byte *wire pixels* (FLAG_WIRE: per frame a plane
of floats -- the bits of a -- followed by a plane of bytes -- the step count, or
255 minus it for a sky pixel; `wire_shape`) and rank 0 expands them while restoring
row order -- the assembled RGBA32F frame is bit for bit the single-GPU frame.

Most of a wire float plane is zeros (sky and unlit pixels), so before the gather a
rank may compact its shares into the *sparse* wire format (`wire_compact`: code
bytes, per 8x8 tile a mask and a slot index, the non-zero floats packed; about
1.2 bytes per pixel + 4 per lit pixel) and rank 0 expands that instead
(`deinterleave_sparse`).  The capacity of the float array is fixed per run: choose
it from a measured maximum (`sparse_count`).  A share with more lit pixels than that
says so in its header (`sparse_headers`): its rank then sends the dense wire share
as well, point to point, and rank 0 writes it over that rank's rows
(`deinterleave_share`) -- the frame is complete either way.
"""
import ctypes

from ._lib import check, lib, need_lab


class BandLayout:
    """Which rows a rank renders, and where they sit in its compact buffer.

    rank0_weight = 1: bands are dealt round-robin (band b -> rank b mod world).
    rank0_weight < 1: rank 0, which also assembles the frame, gets that fraction of a peer's
    share; bands are dealt by largest remaining credit, so every rank's bands stay spread over
    the whole frame (sky and object rows in every share).  A rank's bands sit in its compact
    buffer in increasing band order either way."""

    def __init__(self, height, world, band_rows=16, rank0_weight=1.0, owner=None):
        if band_rows < 1 or world < 1 or height < 1:
            raise ValueError("BandLayout: height, world and band_rows must be positive")
        if not 0.0 < rank0_weight <= 1.0:
            raise ValueError("BandLayout: rank0_weight must be in (0, 1]")
        # any band height renders correctly; multiples of 8 keep whole 8x8 wave tiles inside one band
        self.height, self.world, self.band_rows = int(height), int(world), int(band_rows)
        self.n_bands = (self.height + self.band_rows - 1) // self.band_rows
        self.rank0_weight = float(rank0_weight) if world > 1 else 1.0
        self.weighted = self.rank0_weight < 1.0
        self.dealt_by_cost = owner is not None
        if owner is not None:
            # an explicit deal (balanced_owner: bands dealt by their measured cost): owner[b] = the rank that renders band b
            owner = [int(o) for o in owner]
            if len(owner) != self.n_bands or self.n_bands > 512 or self.world > 64 or any(not 0 <= o < self.world for o in owner):
                raise ValueError("BandLayout: owner must name a rank for each of the frame's bands (at most 512 bands, 64 ranks)")
            if any(r not in owner for r in range(self.world)):
                raise ValueError("BandLayout: every rank must own a band (its share may not be empty)")
            self.owner, self.weighted = owner, True
        elif self.weighted:
            if self.n_bands > 512 or self.world > 64:
                raise ValueError("BandLayout: a weighted layout holds at most 512 bands and 64 ranks")
            # integer credits (no float accumulation: every rank must compute the same deal)
            unit = 1 << 20
            w = [int(round(self.rank0_weight * unit))] + [unit] * (self.world - 1)
            total = sum(w)
            credit = [0] * self.world
            self.owner = []
            for _ in range(self.n_bands):
                for r in range(self.world):
                    credit[r] += w[r]
                best = max(range(self.world), key=lambda r: (credit[r], -r))
                credit[best] -= total
                self.owner.append(best)
        else:
            self.owner = [b % self.world for b in range(self.n_bands)]
        self._bands = [[b for b in range(self.n_bands) if self.owner[b] == r] for r in range(self.world)]
        self._local = [0] * self.n_bands
        for blist in self._bands:
            for lb, b in enumerate(blist):
                self._local[b] = lb
        self.bands_per_rank = max(len(b) for b in self._bands)
        self.rows_per_rank = self.bands_per_rank * self.band_rows   # equal message sizes

    def bands_of(self, rank):
        return list(self._bands[rank])

    def rows_of(self, rank):
        """[(local_row, global_row)] for every real row of `rank`."""
        out = []
        for lb, b in enumerate(self._bands[rank]):
            for r in range(self.band_rows):
                y = b * self.band_rows + r
                if y < self.height:
                    out.append((lb * self.band_rows + r, y))
        return out

    def source_of(self, y):
        """global row -> (rank, local_row): the inverse used by the de-interleave."""
        b = y // self.band_rows
        return self.owner[b], self._local[b] * self.band_rows + y % self.band_rows


def balanced_owner(costs, world, extra0=0.0, max_bands=None):
    """Deal the frame's bands to the ranks by COST (longest processing time first: the most expensive band still undealt goes to
    the rank with the least work so far) instead of by count.  costs[b] = the measured cost of band b (band_costs), extra0 = the
    work rank 0 has besides its bands, in the same unit (it assembles the frame).  No rank receives more than max_bands bands (the
    shares travel in messages of equal size, as large as the largest share: default the even count + a quarter, at least + 2),
    every rank at least one.  Deterministic: every rank that computes it gets the same deal.  -> owner[b]."""
    n = len(costs)
    if world < 1 or n < world:
        raise ValueError("balanced_owner: fewer bands than ranks")
    even = (n + world - 1) // world
    if max_bands is None:
        max_bands = even + max(2, even // 4)
    max_bands = max(max_bands, even)
    load = [float(extra0)] + [0.0] * (world - 1)
    count = [0] * world
    owner = [-1] * n
    order = sorted(range(n), key=lambda b: (-float(costs[b]), b))
    for i, b in enumerate(order):
        left = n - i                                        # bands still to deal, this one included
        empty = [r for r in range(world) if count[r] == 0]
        if len(empty) >= left:                              # the last bands go to the ranks that have none
            r = min(empty, key=lambda r: (load[r], r))
        else:
            r = min((r for r in range(world) if count[r] < max_bands), key=lambda r: (load[r], r))
        owner[b] = r
        load[r] += float(costs[b])
        count[r] += 1
    return owner


def band_costs(frame_alpha, band_rows, fixed=4.0):
    """The cost of every band of a frame from its rendered step counts (the alpha channel of the RGBA32F frame, a torch tensor
    H x W): a wave renders an 8x8 tile and runs as many iterations as its longest pixel takes, so a tile costs max(alpha) + `fixed`
    (the wave's start and its stores, in iterations) and a band the sum over its tiles.  -> list of floats, one per band."""
    import torch
    H, W = frame_alpha.shape
    a = torch.nan_to_num(frame_alpha.float(), nan=0.0, posinf=140.0, neginf=0.0)
    ph, pw = (-H) % 8, (-W) % 8
    if ph or pw:
        a = torch.nn.functional.pad(a, (0, pw, 0, ph))
    t = a.reshape(a.shape[0] // 8, 8, a.shape[1] // 8, 8).amax(dim=(1, 3)) + float(fixed)      # per tile
    rows = t.sum(dim=1)                                                                         # per tile row
    per_band = max(1, band_rows // 8)
    n_bands = (H + band_rows - 1) // band_rows
    out = []
    for b in range(n_bands):
        out.append(float(rows[b * per_band:(b + 1) * per_band].sum().item()) if band_rows % 8 == 0 else
                   float(rows[(b * band_rows) // 8:((b + 1) * band_rows + 7) // 8].sum().item()))
    return out


def group_plan(world, steps=0):
    """(frames per launch and gather, groups in flight, shares launched in tile order) of a sharded run of `steps` frames over
    `world` ranks (0 = a long run).  A rank's share of one frame is small (1 / world of it) behind a fixed cost per launch,
    collective and wait, so G frames share a launch; and a launch alone ends with its longest waves' chain of dependent steps,
    so several are kept in flight.  A short run (the driver's scaling run times 20 steps) is one fill and one drain with little
    between them: its launches take their tiles in the order of their cost in the last launch (SDFHIP_FLAG_TILE_ORDER: the
    longest waves start first; 1080p at 8 ranks: 15.4 -> 14.6 us per frame of a 20-step burst).  A long run keeps the default
    order: the order costs every wave a lookup in front of everything else and launches a group's expensive tiles all at once --
    12.3 -> 13.7 us per frame in the steady state (scripts/rank_emulation.py, profiles/r05_rank_emulation.txt)."""
    G = 8 if world >= 8 else 4
    nbuf = 4
    return G, nbuf, is_burst(world, steps)


def is_burst(world, steps):
    """THE predicate of a short sharded run: `steps` frames are at most four fills of the pipeline (4 x frames per
    launch x launches in flight).  Everything that treats a short run differently asks this one function: group_plan (tile order
    on the batched launches) and bench.py's band deal (timed as that very burst instead of in the steady state).  The line prints
    it (config.plan), and a steady-state figure beside the burst's (`steady_state`)."""
    G = 8 if world >= 8 else 4
    return 0 < steps <= 4 * G * 4


def wire_shape(rows, width):
    """Shape of the uint8 tensor that holds one frame-share of `rows` x `width` wire pixels: the
    first 4 * rows * width bytes are the float plane, the last rows * width the byte plane."""
    if (rows * width) % 4:
        raise ValueError("wire buffers need rows * width to be a multiple of 4")
    return (5, rows, width)


def render_bands(scene, state, width, layout, rank, out_ptr, flags=0, stream=None, stats=None, pt=None):
    """Render `rank`'s bands of the frame into its compact device buffer (pt: path-traced mode)."""
    if layout.weighted:
        scene.DrawBandsDevice([state], width, layout.height, out_ptr, layout.band_rows, layout.bands_of(rank),
                              nrows_out=layout.rows_per_rank, pt=pt, flags=flags, stream=stream, stats=stats)
        return
    kw = dict(nrows_out=layout.rows_per_rank, band_rows=layout.band_rows, band_first=rank,
              band_stride=layout.world, flags=flags, stream=stream, stats=stats)
    if pt is None:
        scene.DrawDevice(state, width, layout.height, out_ptr, **kw)
    else:
        scene.DrawPathDevice(state, width, layout.height, out_ptr, pt=pt, **kw)


def render_bands_batch(scene, states, width, layout, rank, out_ptr, flags=0, stream=None, stats=None):
    """`rank`'s bands of several frames (one camera each) in ONE launch, into
    out_ptr[frame][rows_per_rank][width]: the frames of a gather group share the serial tail of
    the share's longest pixels."""
    if layout.weighted:
        scene.DrawBandsDevice(list(states), width, layout.height, out_ptr, layout.band_rows, layout.bands_of(rank),
                              nrows_out=layout.rows_per_rank, flags=flags, stream=stream, stats=stats)
        return
    scene.DrawBatchDevice(states, width, layout.height, out_ptr, nrows_out=layout.rows_per_rank,
                          band_rows=layout.band_rows, band_first=rank, band_stride=layout.world,
                          flags=flags, stream=stream, stats=stats)


def deinterleave(device, gathered_ptr, frame_ptr, width, layout, stream=None, pixel_bytes=16, frames=1):
    """Rank 0: gathered compact buffers (world x frames x rows_per_rank x width pixels) ->
    frames x height x width.  pixel_bytes: 16 for RGBA32F frames, 4 for RGBA8 frames of the
    fused display pass, 5 for wire buffers (FLAG_WIRE renders: per rank and frame a `wire_shape`
    block), which are expanded to the RGBA32F frame on the way.  frames > 1: one gather carried several frames."""
    st = ctypes.c_void_p(int(stream)) if stream else None
    if layout.weighted:
        owner = (ctypes.c_uint8 * layout.n_bands)(*layout.owner)
        check(lib.sdfhip_deinterleave_bands_device(int(device), ctypes.c_void_p(int(gathered_ptr)),
                                                   ctypes.c_void_p(int(frame_ptr)), int(width), layout.height,
                                                   layout.band_rows, layout.world, layout.rows_per_rank, owner,
                                                   int(pixel_bytes), int(frames), st))
        return
    check(lib.sdfhip_deinterleave_device(int(device), ctypes.c_void_p(int(gathered_ptr)),
                                         ctypes.c_void_p(int(frame_ptr)), int(width),
                                         layout.height, layout.band_rows, layout.world,
                                         layout.rows_per_rank, int(pixel_bytes), int(frames), st))


def sparse_share_bytes(rows, width, capacity):
    """Bytes of one frame-share in the sparse wire format."""
    need_lab("sparse_share_bytes")
    return int(lib.sdfhip_wire_sparse_bytes(int(width), int(rows), int(capacity)))


def wire_compact(device, wire_ptr, sparse_ptr, width, rows, frames, capacity, stream=None):
    """[frames] dense wire shares (FLAG_WIRE renders) -> [frames] sparse shares, on `stream`."""
    need_lab("wire_compact")
    check(lib.sdfhip_wire_compact_device(int(device), ctypes.c_void_p(int(wire_ptr)), ctypes.c_void_p(int(sparse_ptr)),
                                         int(width), int(rows), int(frames), int(capacity),
                                         ctypes.c_void_p(int(stream)) if stream else None))


def sparse_head_offset(rows, width, capacity):
    """Byte offset of a sparse share's header {uint32 lit pixels, uint32 overflowed, 0, 0}."""
    need_lab("sparse_head_offset")
    return int(lib.sdfhip_wire_sparse_head_offset(int(width), int(rows), int(capacity)))


def sparse_headers(sparse_tensor, rows, width, capacity):
    """The headers of the sparse shares in a uint8 tensor [..., sparse_share_bytes], as a strided uint8 view
    [..., 8] (lit pixels, overflowed) -- to be copied to pinned host memory behind the stream that made them."""
    off = sparse_head_offset(rows, width, capacity)
    return sparse_tensor[..., off:off + 8]


def deinterleave_share(device, share_ptr, frame_ptr, width, layout, rank, stream=None, pixel_bytes=5, frames=1):
    """Rank 0: write ONE rank's buffers ([frames] x rows_per_rank x width pixels; pixel_bytes as for
    `deinterleave`) over that rank's rows of the frames: the dense resend of a share whose sparse form
    overflowed its capacity."""
    need_lab("deinterleave_share")
    owner = (ctypes.c_uint8 * layout.n_bands)(*layout.owner) if layout.weighted else None
    check(lib.sdfhip_deinterleave_share_device(int(device), ctypes.c_void_p(int(share_ptr)), ctypes.c_void_p(int(frame_ptr)),
                                               int(width), layout.height, layout.band_rows, layout.world,
                                               layout.rows_per_rank, owner, int(rank), int(pixel_bytes), int(frames),
                                               ctypes.c_void_p(int(stream)) if stream else None))


def sparse_count(sparse_tensor, rows, width, capacity):
    """Lit pixels (float slots needed) and overflow flag of every sparse share in a uint8 tensor
    [..., sparse_share_bytes]: -> (counts, overflowed) as flat lists.  Synchronises."""
    import torch
    nbytes = sparse_share_bytes(rows, width, capacity)
    off_head = sparse_head_offset(rows, width, capacity)
    flat = sparse_tensor.reshape(-1, nbytes)
    head = flat[:, off_head:off_head + 8].contiguous().cpu().view(torch.int32)
    return head[:, 0].tolist(), head[:, 1].tolist()


def deinterleave_sparse(device, gathered_ptr, frame_ptr, width, layout, capacity, stream=None, frames=1, overflow_ptr=None):
    """Rank 0: gathered sparse shares (world x frames x sparse_share_bytes) -> frames x height x width
    RGBA32F.  overflow_ptr: a device int32 that is OR-ed with 1 if a share had more lit pixels than
    `capacity` (the frame is then incomplete)."""
    need_lab("deinterleave_sparse")
    owner = (ctypes.c_uint8 * layout.n_bands)(*layout.owner) if layout.weighted else None
    check(lib.sdfhip_deinterleave_sparse_device(int(device), ctypes.c_void_p(int(gathered_ptr)),
                                                ctypes.c_void_p(int(frame_ptr)), int(width), layout.height,
                                                layout.band_rows, layout.world, layout.rows_per_rank, owner,
                                                int(capacity), int(frames),
                                                ctypes.c_void_p(int(overflow_ptr)) if overflow_ptr else None,
                                                ctypes.c_void_p(int(stream)) if stream else None))


# ---- sparse shares written by the march kernel itself (sdfhip_render_sparse_device) ----------------------------
def sparse2_bytes(rows, width, frames, capacity):
    """Bytes of one share that holds `frames` frames of `rows` x `width` pixels and `capacity` packed floats."""
    return int(lib.sdfhip_sparse2_bytes(int(width), int(rows), int(frames), int(capacity)))


def sparse2_floats_offset(rows, width, frames):
    """Byte offset of a share's float array = the bytes of its fixed part (header, masks, slot bases, codes)."""
    return int(lib.sdfhip_sparse2_floats_offset(int(width), int(rows), int(frames)))


def render_sparse2(scene, states, width, layout, rank, share_ptr, capacity, count_base, flags=0, stream=None):
    """`rank`'s bands of the frames of a group (one camera each) in ONE launch of the default kernel, which writes the sparse
    share itself: no dense wire buffer, no compaction kernels.  count_base = the value of the share's counter (its first
    word) before this launch: the library never zeroes it."""
    from ._lib import Info
    if not isinstance(states, (list, tuple)):
        states = [states]
    infos = (Info * len(states))(*[s if isinstance(s, Info) else s.State for s in states])
    blist = layout.bands_of(rank)
    bands = (ctypes.c_uint16 * len(blist))(*blist)
    check(lib.sdfhip_render_sparse_device(scene._h, infos, len(states), int(width), layout.height, layout.band_rows, bands, len(blist),
                                          layout.rows_per_rank, int(capacity), int(count_base) & 0xFFFFFFFF, int(flags),
                                          ctypes.c_void_p(int(share_ptr)), ctypes.c_void_p(int(stream)) if stream else None))


class SparseShareCall:
    """render_sparse2 for a renderer that launches the same share frame after frame: the band list, the camera array and the
    argument objects are made ONCE; a call copies the group's camera blocks (112 bytes each) and enters the library.  A group's
    launch is then a few microseconds of host time instead of several tens (a rank's share of a frame is 10 - 40 us of GPU work:
    the host must not be what the GPU waits for)."""

    def __init__(self, scene, width, layout, rank, capacity, max_frames=8, flags=0, bands=None):
        from ._lib import Info
        self._h, self._w, self._h_frame = scene._h, int(width), layout.height
        blist = layout.bands_of(rank) if bands is None else list(bands)
        self._bands = (ctypes.c_uint16 * len(blist))(*blist)
        self._nb, self._band_rows, self._rows = len(blist), layout.band_rows, layout.rows_per_rank
        self._cap, self._flags = int(capacity), int(flags)
        self._infos = (Info * max_frames)()
        self._max = max_frames
        self._size = ctypes.sizeof(Info)

    def __call__(self, states, share_ptr, count_base, stream=None, flags=None):
        n = len(states)
        if n > self._max:
            raise ValueError("SparseShareCall: more frames than it was made for")
        for i, st in enumerate(states):              # (always: a camera may have moved in place since the last call; 112 bytes each)
            ctypes.memmove(ctypes.addressof(self._infos) + i * self._size, ctypes.addressof(getattr(st, "State", st)), self._size)
        check(lib.sdfhip_render_sparse_device(self._h, self._infos, n, self._w, self._h_frame, self._band_rows, self._bands, self._nb,
                                              self._rows, self._cap, int(count_base) & 0xFFFFFFFF, self._flags if flags is None else int(flags),
                                              ctypes.c_void_p(int(share_ptr)), ctypes.c_void_p(int(stream)) if stream else None))


class SparseExpandCall:
    """deinterleave_sparse2 with its argument arrays made once (see SparseShareCall)."""

    def __init__(self, device, width, layout, capacity, flags=0):
        self._dev, self._w, self._lay, self._cap, self._flags = int(device), int(width), layout, int(capacity), int(flags)
        self._owner = (ctypes.c_uint8 * layout.n_bands)(*layout.owner) if layout.weighted else None
        self._ptrs = {}

    def __call__(self, key, share_ptrs, frame_ptr, frames=1, only_rank=-1, counts_ptr=None, stream=None):
        """key: names the list of share pointers (a slot number): its array is kept"""
        lay = self._lay
        if key not in self._ptrs:
            self._ptrs[key] = (ctypes.c_void_p * lay.world)(*[int(p) if p else None for p in share_ptrs])
        check(lib.sdfhip_deinterleave_sparse2_device(self._dev, self._ptrs[key], ctypes.c_void_p(int(frame_ptr)), self._w, lay.height,
                                                     lay.band_rows, lay.world, lay.rows_per_rank, self._owner, self._cap, int(frames),
                                                     self._flags, int(only_rank), ctypes.c_void_p(int(counts_ptr)) if counts_ptr else None,
                                                     ctypes.c_void_p(int(stream)) if stream else None))


def deinterleave_sparse2(device, share_ptrs, frame_ptr, width, layout, capacity, frames=1, flags=0, only_rank=-1, counts_ptr=None,
                         stream=None):
    """Rank 0: the ranks' shares (one device pointer per rank; rank 0's own may be the buffer it rendered into) ->
    frames x height x width RGBA32F (RGBA8 with FLAG_DISPLAY[_DEBUG] in `flags`).  counts_ptr: device-accessible memory
    (pinned host memory will do) for the `world` counters of the shares."""
    ptrs = (ctypes.c_void_p * layout.world)(*[int(p) if p else None for p in share_ptrs])
    owner = (ctypes.c_uint8 * layout.n_bands)(*layout.owner) if layout.weighted else None
    check(lib.sdfhip_deinterleave_sparse2_device(int(device), ptrs, ctypes.c_void_p(int(frame_ptr)), int(width), layout.height,
                                                 layout.band_rows, layout.world, layout.rows_per_rank, owner, int(capacity), int(frames),
                                                 int(flags), int(only_rank), ctypes.c_void_p(int(counts_ptr)) if counts_ptr else None,
                                                 ctypes.c_void_p(int(stream)) if stream else None))
