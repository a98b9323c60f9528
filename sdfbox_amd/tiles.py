"""Screen-tile sharding of one frame over the GPUs of a node (new design, the
reference is single-GPU: SURVEY.md 8e).

Pixels are independent and the scene is read-only, so the octree is replicated
on every device and the *frame* is sharded: row bands of `band_rows` rows are
dealt round-robin (band b -> rank b mod world), which balances sky against
object rows.  Each rank renders its bands into a compact local buffer of
`rows_per_rank` rows; one gather per frame brings the buffers to rank 0 over
xGMI (RCCL send/recv: every peer has its own direct link into rank 0), where
a small kernel restores row order.  No other collective is on the data path.
"""
import ctypes

from ._lib import check, lib


class BandLayout:
    """Which rows a rank renders, and where they sit in its compact buffer."""

    def __init__(self, height, world, band_rows=16):
        if band_rows < 1 or world < 1 or height < 1:
            raise ValueError("BandLayout: height, world and band_rows must be positive")
        # any band height renders correctly; multiples of 16 keep whole workgroup
        # tiles (16x16 pixels) inside one band
        self.height, self.world, self.band_rows = int(height), int(world), int(band_rows)
        self.n_bands = (self.height + self.band_rows - 1) // self.band_rows
        self.bands_per_rank = (self.n_bands + self.world - 1) // self.world
        self.rows_per_rank = self.bands_per_rank * self.band_rows   # equal message sizes

    def bands_of(self, rank):
        return list(range(rank, self.n_bands, self.world))

    def rows_of(self, rank):
        """[(local_row, global_row)] for every real row of `rank`."""
        out = []
        for lb, b in enumerate(self.bands_of(rank)):
            for r in range(self.band_rows):
                y = b * self.band_rows + r
                if y < self.height:
                    out.append((lb * self.band_rows + r, y))
        return out

    def source_of(self, y):
        """global row -> (rank, local_row): the inverse used by the de-interleave."""
        b = y // self.band_rows
        return b % self.world, (b // self.world) * self.band_rows + y % self.band_rows


def render_bands(scene, state, width, layout, rank, out_ptr, flags=0, stream=None, stats=None, pt=None):
    """Render `rank`'s bands of the frame into its compact device buffer (pt: path-traced mode)."""
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
    scene.DrawBatchDevice(states, width, layout.height, out_ptr, nrows_out=layout.rows_per_rank,
                          band_rows=layout.band_rows, band_first=rank, band_stride=layout.world,
                          flags=flags, stream=stream, stats=stats)


def deinterleave(device, gathered_ptr, frame_ptr, width, layout, stream=None, pixel_bytes=16, frames=1):
    """Rank 0: gathered compact buffers (world x frames x rows_per_rank x width pixels) ->
    frames x height x width.  pixel_bytes: 16 for RGBA32F frames, 4 for RGBA8 frames of the
    fused display pass.  frames > 1: one gather carried several frames."""
    check(lib.sdfhip_deinterleave_device(int(device), ctypes.c_void_p(int(gathered_ptr)),
                                         ctypes.c_void_p(int(frame_ptr)), int(width),
                                         layout.height, layout.band_rows, layout.world,
                                         layout.rows_per_rank, int(pixel_bytes), int(frames),
                                         ctypes.c_void_p(int(stream)) if stream else None))
