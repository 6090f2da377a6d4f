// splat2.hpp - schedule-driven owner-computes push (splat2.hip).
//
// The affine of an operator is constant for thousands of matvecs, so everything about
// "which grid points land in which output tile" is computed ONCE per operator by a build
// kernel (the splat schedule) and the hot kernel only streams that list:
//   per output tile (8 x 4 x 30, aproned accumulator in LDS, one wave per tile)
//     a list of 64-lane instructions; an instruction packs up to 8 segments of grid rows
//     (1..32 consecutive grid-z points each) whose points ALL land in the tile's aproned cell
//     range and the field of view - decided exactly, with the kernel's own float arithmetic -
//     cut where two consecutive points share a z plane, and packed only with segments whose
//     footprints provably cannot meet.
// The hot loop therefore has no bounds tests, no row enumeration, no conflict detection and no
// lane hand-over: decode (segment-start bitmask -> mbcnt -> one 16-byte LDS read), 3 FMAs,
// floor/fract, 14 products and two dense 4-cell read-add-write groups (z plane, then z+1
// plane) per 64 lanes.
#pragma once
#include "fused.hpp"

namespace unires {

constexpr unsigned kS2RowBits = 19, kS2RowIdle = (1u << kS2RowBits) - 1u;

struct S2Entry {     // one segment of an instruction (16 bytes)
  float rx, ry, rz;  // affine_row(A, ui, uj): the row part of the coordinate arithmetic
  unsigned pk;       // row code (19 bits) | (k0 - first lane + 64) << 19; axis 2 / 3: k0 relative to the tile's table base
};

struct S2Ext {  // axis 3 (conv_up along all three axes): per-segment x / y part of the conv_up
  unsigned base4;          // byte offset of x-space voxel (kx, ky, 0)
  float w00, w01, w10, w11;  // wx_a * wy_b for the 2 x 2 x-space columns that feed the row
  unsigned pad[3];
};

struct SplatSched {
  S2Entry *entries = nullptr;            // device
  S2Ext *ext = nullptr;                  // device, axis 3 only (same indexing as entries)
  ulonglong2 *masks = nullptr;           // device, per instruction: {bit l-1 set <=> a segment starts at lane l, active lanes}
  uint2 *tile_off = nullptr;             // device, ntiles + 1 {entry offset, instruction offset}, in PROCESSING order (build kernels)
  int *tile_geom = nullptr;              // device, ntiles: the output tile of processing slot u (build kernels)
  // device, one 16-byte record per position of the walk (splat2_build (4)): {tile index x | y << 10 | z << 20,
  // entry offset, instruction offset, instructions | conv_up table base << 8}
  uint4 *recs = nullptr;
  size_t cap_recs = 0;
  int pos_lo[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};  // position range of each partition (XCD): contiguous tiles of equal COST
  int nwg = 0;                            // workgroups the layout was made for (the launch must use as many)
  unsigned long long *scratch = nullptr; // device: {error flag, points, instructions} of a build
  size_t cap_entries = 0, cap_instr = 0, cap_tiles = 0;
  int ntiles = 0;
  bool valid = false;
  int axis = -1;      // -1 direct source; 0..2 conv_up along that axis; 3 along all three
  double fill = 0.0;  // active lanes / issued lanes (diagnostic)
};

// Build (or rebuild) the schedule of one operator.  Synchronises the device (plan-time only).
// Row code of a segment: axis 2 / -1: ui * rows_y + uj (the source row index); axis 0 / 1:
// ui << 9 | uj.  Returns non-zero if the operator is outside the kernel's domain (schedule left
// invalid; callers use the general kernels).
// axis 3: xtab / ytab = device conv_up tables (splat2_convtab) along x and y, xd = x-space dims.
int splat2_build(SplatSched &S, const Affine &A, const Affine &Ainv, Dim3i gd, Dim3i dd, float tol,
                 const SplatSafety &safe, int axis, int rows_y, const float4 *xtab = nullptr,
                 const float4 *ytab = nullptr, Dim3i xd = Dim3i{0, 0, 0});
void splat2_free(SplatSched &S);

// How hard the schedule builders (splat2_build, ata1_build) of the calling thread try: thorough = rows closer than
// row_sep are tested point by point (a better lane fill, 2 - 4 x the build time); quick = kept apart wholesale.
// The plan builds thoroughly once (unires_plan_create) and quickly whenever an operator changes under a running
// reconstruction (unires_plan_set_repeat: every rigid Gauss-Newton step).
void sched_set_thorough(bool on);
bool sched_thorough();

int splat2_blocks(Dim3i dd, int grid_cap = 0);  // (grid_cap: PushEpilogue::grid_cap of the launch)
// tab_dev: gn float4 {bits(koff), w0, w1, -} conv_up table along the schedule's axis (nullptr for
// a direct source); row_stride: elements per source row (axis 2 / -1), per ui (axis 1) or per uj
// (axis 0); tab_step: elements between the two x-space values of a grid voxel (axis 0 / 1).
// Non-zero return: nothing launched.
int launch_splat2(const SplatSched &S, const float *src, size_t src_numel, const float4 *tab_dev, int gn,
                  unsigned row_stride, unsigned tab_step, unsigned xs_sy, unsigned xs_sx, const Affine &A,
                  float alpha,
                  const PushEpilogue &ep, float *dst, Dim3i dd, const int *done, hipStream_t st);

// host: conv_up table along `axis` (gn entries of 4 floats), {bits(first x-space index), w0, w1, -}
void splat2_convtab(const Taps &T, const Scaling &S, int axis, int gn, int xdn, float *out);

}  // namespace unires
