"""raytracers_amd -- MI355X-native (gfx950) render hot path of athas/raytracers.

The product is libray_mi355x.so (hand-written HIP kernels behind a C ABI; see include/).
This package is the thin host-side mirror of the reference's call surface plus the
row-tile sharding used for multi-GPU renders.  Importing it loads the HIP library and
fails loudly if it has not been built."""
from .api import (Context, Prepared, RtError, Scene, MAX_DEPTH, ROWS_PER_TILE, VARIANT_AUTO, VARIANT_PERSISTENT, VARIANT_POOLED,  # noqa: F401
                  VARIANT_PIXEL, part_rows, place_part, place_parts, place_parts_batch, prepare_scene, render, render_batch_into, render_image, render_inplace_into, render_into, render_timed,
                  ipc_export, ipc_import, ipc_close)
