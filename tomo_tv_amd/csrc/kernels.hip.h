// kernels.hip.h -- gfx950 device kernels of the tomography hot path.
//
// Device layout ("slab-interleaved"): every field is stored with the slice index
// fastest,   volume  x[pixel = y*N + z][s]   and   sinogram  g[row = angle*N + ray][s],   row pitch sx floats
// (sx = Nslice rounded up to 64, padding slices are identically zero).  All slices share one system
// matrix, so a matrix entry (row, pixel, w) is wave-uniform scalar data and its use is one coalesced
// vector AXPY across slices: lanes index slices, the scalar unit walks the ray / voxel tables.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// The kernels live in one header per family; the order matters (constants and helpers are shared downwards):
#include "kernels_common.hip.h"   // vector helpers, reductions, transposes, the register-block typedefs
#include "kernels_fp.hip.h"       // forward projectors (defines row_ror, the FT_ tile shape)
#include "kernels_bp.hip.h"       // back projectors (defines CellD; k_bp_tile reuses the FT_ tile shape)
#include "kernels_sart.hip.h"     // streamed SART / ART sweeps (use CellD, row_ror, FT_BATCH)
#include "kernels_misc.hip.h"     // element-wise passes, reductions, multimodal, CGLS scalars, WBP filter
#include "kernels_tv.hip.h"       // TV value / gradient / update
#include "kernels_fgp.hip.h"      // FGP-TV
