// splat2.hpp - schedule-driven owner-computes push (splat2.hip).
//
// The affine of an operator is constant for thousands of matvecs, so everything about
// "which grid points land in which output tile" is computed ONCE per operator by a build
// kernel (the splat schedule) and the hot kernel only streams that list:
//   per output tile (8 x 4 x 30, aproned accumulator in LDS, one wave per tile)
//     a list of instructions = pairs of <= 32-point segments of grid rows whose points ALL
//     land in the tile's aproned cell range and the field of view (decided exactly, with the
//     kernel's own float arithmetic), cut where two consecutive points share a z plane, and
//     paired only with rows whose footprints provably cannot meet.
// The hot loop therefore has no bounds tests, no row enumeration, no conflict detection and no
// lane hand-over: decode, 3 FMAs, floor/fract, 14 products and two dense 4-cell
// read-add-write groups (z plane, then z+1 plane) per 64 lanes.
#pragma once
#include "fused.hpp"

namespace unires {

struct S2Entry {   // one segment of an instruction (24 bytes)
  float rx, ry, rz;  // affine_row(A, ui, uj): the row part of the coordinate arithmetic
  float k0f;         // first grid z of the segment
  unsigned srcoff;   // element offset of the row in the source volume (ui * sx + uj * sy)
  unsigned kl;       // k0 | len << 12 | table index << 18
};

struct SplatSched {
  S2Entry *entries = nullptr;    // device
  unsigned *tile_off = nullptr;  // device, ntiles + 1 entry offsets (2 entries per instruction)
  unsigned long long *scratch = nullptr;  // device: {error flag, points, instructions} of a build
  size_t cap_entries = 0, cap_tiles = 0;
  int ntiles = 0;
  unsigned total = 0;  // entries in use
  bool valid = false;
  int axis = -1;       // -1 direct source; 0..2 conv_up along that axis
  double fill = 0.0;   // active lanes / issued lanes (diagnostic)
};

// Build (or rebuild) the schedule of one operator.  Synchronises the device (plan-time only).
// sx, sy: source row strides in elements; tabsel: which row index selects the conv table entry
// (0: ui, 1: uj; unused for axis 2 / -1).  Returns non-zero if the operator is outside the
// kernel's domain (schedule left invalid; callers use the general kernels).
int splat2_build(SplatSched &S, const Affine &A, const Affine &Ainv, Dim3i gd, Dim3i dd, float tol,
                 const SplatSafety &safe, int axis, unsigned sx, unsigned sy);
void splat2_free(SplatSched &S);

int splat2_blocks(Dim3i dd);
// tab_dev: gn float4 {bits(koff), w0, w1, -} conv_up table along the schedule's axis (nullptr for
// a direct source).  Non-zero return: nothing launched.
int launch_splat2(const SplatSched &S, const float *src, size_t src_numel, const float4 *tab_dev, int gn,
                  unsigned tab_step,
                  const Affine &A, float alpha, float tol, const PushEpilogue &ep, float *dst,
                  Dim3i dd, const int *done, hipStream_t st);

// host: conv_up table along `axis` (gn entries of 4 floats), same packing as gather2_ztab
void splat2_convtab(const Taps &T, const Scaling &S, int axis, int gn, int xdn, float *out);

}  // namespace unires
