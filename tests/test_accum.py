"""wr_accum (csrc/wrhip_k_setup.h): the n-fold sequential fp32 sum s0 + step + step + ... that swgl's
span loops and Edge::nextRow produce by repeated addition, evaluated binade by binade.  The kernel
header is compiled for the host here and checked against the plain loop, bit for bit, on random,
dyadic, tie-prone, binade-floor and raw-bit-pattern inputs."""
import os
import subprocess
import sys
import pytest
from conftest import ROOT

SRC = r'''
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#define WRHIP_HOSTSIM 1
#define WR_DEVICE static inline
#define __device__
#define __global__
#define __noinline__
#define __launch_bounds__(x)
#include "wrhip_types.h"
static float wr_low_bit_dummy;
#include "accum_only.h"
static float ref(float s, float d, int c) { for (int i = 0; i < c; i++) s += d; return s; }
static uint64_t rs = 88172645463325252ull;
static uint64_t rnd() { rs ^= rs << 13; rs ^= rs >> 7; rs ^= rs << 17; return rs; }
static float rf(float lo, float hi) { return lo + (hi - lo) * (float)((rnd() >> 11) * (1.0 / 9007199254740992.0)); }
int main() {
  long bad = 0, n = 0;
  for (int it = 0; it < 1600000; it++) {
    float s, d; int c = (int)(rnd() % 3000);
    switch (it % 8) {
      case 0: s = rf(0, 1); d = rf(-0.01f, 0.01f); break;
      case 1: s = rf(-1, 1); d = rf(-0.001f, 0.001f); break;
      case 2: s = rf(0, 2000); d = rf(-2, 2); break;
      case 3: s = 0.0f; d = rf(-0.01f, 0.01f); break;
      case 4: s = rf(0, 1); d = ldexpf((float)(rnd() % 7) - 3.0f, -(int)(rnd() % 30)); break;
      case 5: s = ldexpf(1.0f, (int)(rnd() % 20) - 10); d = -rf(0, 1) * ldexpf(1.0f, (int)(rnd() % 30) - 28); break;
      case 6: { uint32_t a = (uint32_t)rnd(), bb = (uint32_t)rnd(); memcpy(&s, &a, 4); memcpy(&d, &bb, 4); if (!isfinite(s) || !isfinite(d)) { s = 1; d = 1; } c %= 200; break; }
      default: s = rf(-1e-3f, 1e-3f); d = rf(0, 1e-3f); break;
    }
    float a = ref(s, d, c), b = wr_accum(s, d, c), e = wr_accum_binades(s, d, c);
    n++;
    if ((memcmp(&a, &b, 4) != 0 && !(a != a && b != b)) || (memcmp(&a, &e, 4) != 0 && !(a != a && e != e))) {
      if (bad < 10) printf("MISMATCH s=%a d=%a c=%d ref=%a accum=%a binades=%a\n", s, d, c, a, b, e);
      bad++;
    }
  }
  printf("%ld cases, %ld mismatches\n", n, bad);
  return bad != 0;
}
'''


def test_accum_matches_sequential_adds(tmp_path):
    hdr = open(os.path.join(ROOT, "webrender_amd", "csrc", "wrhip_k_setup.h")).read()
    a = hdr.index("WR_DEVICE int wr_low_bit_exp(float x)")
    b = hdr.index("// row-k edge interpolant")
    (tmp_path / "accum_only.h").write_text("#define WR_DBG_PATH(i) ((void)0)\n" + hdr[a:b])
    (tmp_path / "t.cpp").write_text(SRC)
    exe = tmp_path / "t"
    subprocess.check_call(["g++", "-O2", "-ffp-contract=off", "-I", os.path.join(ROOT, "webrender_amd", "csrc"),
                           "-o", str(exe), str(tmp_path / "t.cpp"), "-lm"])
    out = subprocess.run([str(exe)], capture_output=True, text=True)
    sys.stdout.write(out.stdout)
    assert out.returncode == 0, out.stdout[-2000:]


TAB_SRC = r'''
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#define WRHIP_HOSTSIM 1
#define WR_DEVICE static inline
#define __device__
#define __global__
#define __noinline__
#define __launch_bounds__(x)
#include "wrhip_types.h"
static inline float wr_bits_f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline uint32_t wr_float_bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
struct wr_u4 { uint32_t x, y, z, w; };
static inline wr_u4 wr_load16(const void* p) { wr_u4 r; memcpy(&r, p, 16); return r; }
#include "accum_only.h"
static uint64_t rs = 1234567891234567ull;
static uint64_t rnd() { rs ^= rs << 13; rs ^= rs >> 7; rs ^= rs << 17; return rs; }
static float rf(float lo, float hi) { return lo + (hi - lo) * (float)((rnd() >> 11) * (1.0 / 9007199254740992.0)); }
int main() {
  long bad = 0, n = 0, tabs = 0, nofit = 0, pieces = 0;
  static WrAccTabs T;
  for (int it = 0; it < 60000; it++) {
    float s, d; int kmax = 1 + (int)(rnd() % 2500);
    switch (it % 8) {
      case 0: s = rf(0, 1); d = rf(-0.01f, 0.01f); break;
      case 1: s = rf(-1, 1); d = rf(-0.001f, 0.001f); break;
      case 2: s = rf(0, 2000); d = rf(-2, 2); break;
      case 3: s = 0.0f; d = rf(-0.01f, 0.01f); break;
      case 4: s = rf(0, 1); d = ldexpf((float)(rnd() % 7) - 3.0f, -(int)(rnd() % 30)); break;
      case 5: s = ldexpf(1.0f, (int)(rnd() % 20) - 10); d = -rf(0, 1) * ldexpf(1.0f, (int)(rnd() % 30) - 28); break;
      case 6: { uint32_t a = (uint32_t)rnd(), bb = (uint32_t)rnd(); memcpy(&s, &a, 4); memcpy(&d, &bb, 4); if (!isfinite(s) || !isfinite(d)) { s = 1; d = 1; } kmax %= 200; kmax++; break; }
      default: s = rf(-1e-3f, 1e-3f); d = rf(0, 1e-3f); break;
    }
    // the pair (i, i + 2) shares a table when the right sum equals the left one; a third of the cases give the right edge its own
    const float s0[8] = {s, 0, (it % 3) ? s : s + 0.25f, 0, 0, 0, 0, 0}, st[8] = {d, 0, d, 0, 0, 0, 0, 0};
    wr_acctabs_build(&T, 4, s0, st, kmax, false);
    if (T.mode[0] == 2) { tabs++; pieces += T.n[0]; } else if (T.mode[0] == 3) nofit++;
    float a0 = s0[0], a2 = s0[2];
    for (int k = 0; k <= kmax; k++) {
      const float b0 = wr_acctabs_row_any(&T, 0, k), b2 = wr_acctabs_row_any(&T, 2, k);
      n += 2;
      if ((memcmp(&a0, &b0, 4) != 0 && !(a0 != a0 && b0 != b0)) || (memcmp(&a2, &b2, 4) != 0 && !(a2 != a2 && b2 != b2))) {
        if (bad < 10) printf("MISMATCH s=%a d=%a k=%d of %d ref=%a / %a got %a / %a (n %d modes %d %d)\n", s, d, k, kmax, a0, a2, b0, b2, T.n[0], T.mode[0], T.mode[2]);
        bad++;
      }
      a0 += d; a2 += d;
    }
  }
  printf("%ld rows, %ld mismatches; %ld tables (%.1f pieces on average), %ld walks that did not fit\n", n, bad, tabs, tabs ? (double)pieces / tabs : 0.0, nofit);
  return bad != 0;
}
'''


def test_row_sum_tables_match_sequential_adds(tmp_path):
    """WrAccTab (the setup stage's per-prim row-sum tables): every row of every table against the plain loop."""
    hdr = open(os.path.join(ROOT, "webrender_amd", "csrc", "wrhip_k_setup.h")).read()
    a = hdr.index("WR_DEVICE int wr_low_bit_exp(float x)")
    b = hdr.index("// round_pixel (portable path)")
    (tmp_path / "accum_only.h").write_text("#define WR_DBG_PATH(i) ((void)0)\n" + hdr[a:b])
    (tmp_path / "t.cpp").write_text(TAB_SRC)
    exe = tmp_path / "t"
    subprocess.check_call(["g++", "-O2", "-ffp-contract=off", "-I", os.path.join(ROOT, "webrender_amd", "csrc"),
                           "-o", str(exe), str(tmp_path / "t.cpp"), "-lm"])
    out = subprocess.run([str(exe)], capture_output=True, text=True)
    sys.stdout.write(out.stdout)
    assert out.returncode == 0, out.stdout[-2000:]
