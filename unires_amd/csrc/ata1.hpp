// ata1.hpp - single-pass A^T A for operators WITHOUT a slice profile (regime 'denoising':
// A = pull_M, A^T A = push_M . pull_M, unires/_project.py:180-188), ata1.hip.
//
// One wave owns a small output tile (TX x TY x 30 cells) and keeps BOTH the part of p the tile's
// grid points can touch (the tile + one cell all round) and the tile's aproned accumulator in LDS.
// A grid point is then visited once per tile it touches: its coordinates, cell index and weights
// are computed once and serve the gather (8 corners of the p window -> v), the scatter (w_i v into
// the same 8 cells of the accumulator) and - through the window - the stencil / dot epilogue.
// No grid- or x-space intermediate exists, no second kernel, no second read of p.
//
// Which grid points a tile visits is fixed by the operator: a per-operator schedule (ata1_build,
// once per plan / set_repeat) lists, per tile, 64-lane instructions of row segments (lane = z plane
// of the point inside each 32-lane half, so LDS banks never collide and lanes of a half never
// share a cell) whose footprints provably cannot meet.  16 bytes per segment and per instruction.
#pragma once
#include "fused.hpp"

namespace unires {

struct F1Sched {
  uint4 *desc = nullptr;       // device: one per segment {rx, ry, rz, pk}
  uint4 *hdr = nullptr;        // device: one per instruction {segment starts lo, hi, first entry of the tile's, segments}
  uint2 *tile_off = nullptr;   // device: ntiles + 1 {first entry, first instruction} of slot u (processing order)
  int *tile_geom = nullptr;    // device: output tile of processing slot u
  uint2 *tile_org = nullptr;   // device: its origin {x0 | y0 << 16, z0}
  unsigned long long *scratch = nullptr;
  size_t cap_entries = 0, cap_instr = 0, cap_tiles = 0;
  int ntiles = 0;
  int tx = 0, ty = 0;          // tile shape the schedule was built for
  int xcd_lo[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  bool valid = false;
  double fill = 0.0;           // active lanes / issued lanes
  double visits = 0.0;         // grid points visited per output voxel
};

// Build (or rebuild) the schedule of one operator; synchronises the device (plan time only).
// Non-zero: the operator is outside the kernel's domain (schedule left invalid).
int ata1_build(F1Sched &S, const Affine &A, const Affine &Ainv, Dim3i gd, Dim3i dd, float tol,
               const SplatSafety &safe);
void ata1_free(F1Sched &S);

int ata1_blocks(Dim3i dd, int grid_cap = 0);  // partials written by a launch (grid_cap: PushEpilogue::grid_cap)
// dst = [dst +] alpha * push_A(pull_A(src)) [+ c DtD p] with ep as in the push kernels (ep.p, when
// given, must be src).  Non-zero return: nothing launched.
int launch_ata1(const F1Sched &S, const float *src, const Affine &A, float alpha, const PushEpilogue &ep,
                float *dst, Dim3i dd, const int *done, hipStream_t st);

}  // namespace unires
