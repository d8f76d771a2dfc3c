// Addressing the STRIP-MAJOR copy of an int4 matrix (aphro_wna16_strip_relayout, wna16_gemm_resident.hip) from kernels that
// walk the [K/8, N] order: every 16-byte piece of that order (one packed row, four 4-aligned columns) is 16 contiguous bytes
// of the strip-major order, at dword
//     ((ky * S + strip) * nwv + wv) * wave_dw + colbase(col) + mult(col) * (256 s + 64 u + 16 g)
// with packed row = 16 * ((ky * nwv + wv) * nseg + s) + 4 g + u; (colbase, mult) = (pass * nseg * 1024 + col % 64, 4) in the
// 64-column passes of a strip, (np4 * nseg * 1024 + col - 64 np4, rem) in its remainder pass (col: column inside the strip).
// The column part is fixed per lane, the row part uniform over the lanes that share a packed row.  (wna16_gemm_large.hip carries
// the same arithmetic in its Wna16LargeParams, scheduled by hand into its K loop.)
#pragma once
#include <cstdint>

extern "C" int aphro_wna16_strip_geometry(int64_t M, int64_t N, int64_t K, int64_t groups, int* geom);

namespace aphro {

struct Wna16StripGeom {
  int on;                                   // 0: the weights are [K/8, N]
  int nwv, nseg, np4, rem, S;
  uint32_t wave_dw;
  uint32_t inv_nseg, inv_nwv;               // ceil(2^16 / d): (x * inv) >> 16 == x / d for every segment index of the matrix (host-checked)
};

// geometry of the copy laid out for the M class strip_m; false: no strip-major form / the multiply-shift division does not hold
static inline bool wna16_strip_fill(Wna16StripGeom& g, int64_t strip_m, int64_t N, int64_t K, int64_t groups) {
  g = Wna16StripGeom{};
  int geom[5];
  if (strip_m <= 0 || aphro_wna16_strip_geometry(strip_m, N, K, groups, geom) != 1) return false;
  g.on = 1; g.nwv = geom[0]; g.nseg = geom[1]; g.np4 = geom[2]; g.rem = geom[3];
  g.S = (int)(N / (64 * g.np4 + 16 * g.rem));
  g.wave_dw = (uint32_t)(g.nseg * 256 * (4 * g.np4 + g.rem));
  g.inv_nseg = (65536u + g.nseg - 1) / g.nseg;
  g.inv_nwv = (65536u + g.nwv - 1) / g.nwv;
  for (uint32_t x = 0; x <= (uint32_t)(K / 128); ++x)
    if (((x * g.inv_nseg) >> 16) != x / g.nseg || ((x * g.inv_nwv) >> 16) != x / g.nwv) return false;
  return true;
}

#ifdef __HIPCC__
// lane part: dword offset of column `col` (4-aligned, absolute) at (ky, wv, s, u, g) = 0, and the row multiplier
__device__ __forceinline__ uint32_t wna16_strip_col(const Wna16StripGeom& g, int col, uint32_t& mult) {
  const int cw = 64 * g.np4 + 16 * g.rem;
  const int strip = col / cw, cin = col - strip * cw;
  const bool p4 = cin < 64 * g.np4;
  mult = p4 ? 4u : (uint32_t)g.rem;
  return (uint32_t)(strip * g.nwv) * g.wave_dw +
         (p4 ? (uint32_t)(cin >> 6) * g.nseg * 1024u + (cin & 63) : (uint32_t)g.np4 * g.nseg * 1024u + (cin - 64 * g.np4));
}
// row part of packed row `row`: chunk = dword offset of its (ky, wv) chunk, R = 256 s + 64 u + 16 g (to be scaled by mult)
__device__ __forceinline__ void wna16_strip_row(const Wna16StripGeom& g, uint32_t row, uint32_t& chunk, uint32_t& R) {
  const uint32_t seg = row >> 4;
  const uint32_t kw = (seg * g.inv_nseg) >> 16, s = seg - kw * (uint32_t)g.nseg;
  const uint32_t ky = (kw * g.inv_nwv) >> 16, wv = kw - ky * (uint32_t)g.nwv;
  chunk = (ky * (uint32_t)(g.S * g.nwv) + wv) * g.wave_dw;
  R = 256u * s + 64u * (row & 3u) + 16u * ((row >> 2) & 3u);
}
#endif

}  // namespace aphro
