// Which 16 x 16 output tiles of an r x r product each of the 8 waves of a workgroup owns (project_fused.hip,
// solve.hip).
#pragma once

// Wave w owns one rectangular block (rows i0..i0+ni-1 x columns j0..j0+nj-1 of the tile grid).
// Waves w and w + 4 share a SIMD; the layouts balance MFMA work per SIMD as far as the tile count allows
// (r = 80: 25 tiles = 7 + 6 + 6 + 6; rotating the heavy SIMD with the workgroup index changed nothing measurable).
struct Blk { int i0, ni, j0, nj; };
template <int TR> struct Layout;
#define NOBLK {0, 0, 0, 0}
template <> struct Layout<1> { static constexpr Blk blk[8] = {{0,1,0,1}, NOBLK, NOBLK, NOBLK, NOBLK, NOBLK, NOBLK, NOBLK}; };
template <> struct Layout<2> { static constexpr Blk blk[8] = {{0,1,0,1}, {0,1,1,1}, {1,1,0,1}, {1,1,1,1}, NOBLK, NOBLK, NOBLK, NOBLK}; };
template <> struct Layout<3> { static constexpr Blk blk[8] = {{0,1,0,3}, {1,1,0,2}, {1,2,2,1}, {2,1,0,2}, NOBLK, NOBLK, NOBLK, NOBLK}; };
template <> struct Layout<4> { static constexpr Blk blk[8] = {{0,1,0,2}, {0,1,2,2}, {1,1,0,2}, {1,1,2,2}, {2,1,0,2}, {2,1,2,2}, {3,1,0,2}, {3,1,2,2}}; };
template <> struct Layout<5> { static constexpr Blk blk[8] = {{0,2,0,2}, {0,2,2,2}, {2,2,0,2}, {2,2,2,2}, {4,1,0,3}, {0,2,4,1}, {2,2,4,1}, {4,1,3,2}}; };
template <> struct Layout<6> { static constexpr Blk blk[8] = {{0,2,0,3}, {0,2,3,3}, {2,2,0,3}, {2,2,3,3}, {4,1,0,3}, {4,1,3,3}, {5,1,0,3}, {5,1,3,3}}; };
template <> struct Layout<7> { static constexpr Blk blk[8] = {{0,3,0,3}, {0,3,3,3}, {3,3,0,3}, {3,3,3,3}, {0,3,6,1}, {3,3,6,1}, {6,1,0,4}, {6,1,4,3}}; };
template <> struct Layout<8> { static constexpr Blk blk[8] = {{0,2,0,4}, {0,2,4,4}, {2,2,0,4}, {2,2,4,4}, {4,2,0,4}, {4,2,4,4}, {6,2,0,4}, {6,2,4,4}}; };
#undef NOBLK

template <int TR>
constexpr bool layout_covers() {
  int seen[8][8] = {};
  for (int w = 0; w < 8; ++w) {
    const Blk b = Layout<TR>::blk[w];
    for (int i = b.i0; i < b.i0 + b.ni; ++i)
      for (int j = b.j0; j < b.j0 + b.nj; ++j) ++seen[i][j];
  }
  for (int i = 0; i < TR; ++i)
    for (int j = 0; j < TR; ++j)
      if (seen[i][j] != 1) return false;
  return true;
}
static_assert(layout_covers<1>() && layout_covers<2>() && layout_covers<3>() && layout_covers<4>() &&
                  layout_covers<5>() && layout_covers<6>() && layout_covers<7>() && layout_covers<8>(),
              "every output tile belongs to exactly one wave");


// the same table at run time (tr = 1..8, wave = 0..7)
__host__ __device__ inline Blk tile_block(int tr, int wave) {
  switch (tr) {
    case 1: return Layout<1>::blk[wave];
    case 2: return Layout<2>::blk[wave];
    case 3: return Layout<3>::blk[wave];
    case 4: return Layout<4>::blk[wave];
    case 5: return Layout<5>::blk[wave];
    case 6: return Layout<6>::blk[wave];
    case 7: return Layout<7>::blk[wave];
    default: return Layout<8>::blk[wave];
  }
}
