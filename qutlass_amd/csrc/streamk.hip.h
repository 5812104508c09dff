// Stream-K unit walk shared by the persistent GEMM kernels (gemm_nvf4_pk.hip.h, gemm_mx_deepp.hip.h): which (tile, K-stage range) units one
// workgroup computes, in order.  Host and device run the same code: the CPU tests replay every workgroup's walk through the debug entries of
// capi.hip and check that each K stage of each tile is computed exactly once and that parked / consumed scratch slots pair up.
// Reference counterpart: the CUTLASS tile scheduler behind qutlass/csrc/gemm.cu:73-75 (an M-bucketed tile choice, gemm.cu:195-222, is all the
// reference adds on top of it).
//
// T tiles, KT K-stages each.  The first T - Tsk tiles are walked whole, round-robin (workgroup w takes tiles w, w + G, ...); the last Tsk tiles
// form ONE stream of Tsk x KT stages that is cut into G contiguous ranges of (almost) equal length.  A tile cut by a range boundary is computed
// by two workgroups:
//   mode 1  the workgroup that owns its LAST stages [kb, KT) runs them FIRST in its walk and parks the raw fp32 accumulators in scratch slot w
//   mode 2  the workgroup that owns its FIRST stages [0, ke) runs them LAST, combines them with the parked part (slot w + 1) and writes D
//   mode 0  whole tile
// Ranges are at least KT long (the planners use the stream form for MORE than one round of tiles only: Tsk > G), so a tile is cut at most once and no
// range is empty -- the slot pairing (a part is parked by the NEXT workgroup of the walk) relies on both; a boundary is moved to the tile edge
// when it would leave a part shorter than `minp` stages, and lies on a multiple of `gran` stages (the MX kernels unroll their stage code by
// LDS-buffer parity: gran = 2).
#pragma once
#include <hip/hip_runtime.h>

namespace qamd {

struct SkUnit { int tile, kb, ke, mode, slot; };

__host__ __device__ inline int sk_bound(int w, int G, long long Wsk, int KT, int minp, int gran) {
  const long long x = ((long long)w * Wsk + G / 2) / G;
  const int t = (int)(x / KT);
  int o = (int)(x % KT);
  o = (o + gran / 2) / gran * gran;
  if (o < minp) o = 0;
  else if (o > KT - minp) o = KT;
  return t * KT + o;
}

struct SkWalk {
  int w, G, KT, Tdp, dp, pos, end;
  __host__ __device__ SkWalk(int w_, int G_, int T, int Tsk, int KT_, int minp, int gran) : w(w_), G(G_), KT(KT_), Tdp(T - Tsk), dp(w_), pos(0), end(0) {
    if (Tsk > 0) {
      const long long Wsk = (long long)Tsk * KT;
      pos = sk_bound(w, G, Wsk, KT, minp, gran);
      end = sk_bound(w + 1, G, Wsk, KT, minp, gran);
    }
  }
  __host__ __device__ SkUnit next() {
    SkUnit u = {0, 0, 0, -1, 0};
    if (dp < Tdp) {   // whole tiles w, w + G, ... of the data-parallel part
      u.tile = dp; u.kb = 0; u.ke = KT; u.mode = 0;
      dp += G;
    } else if (pos < end) {
      const int t = pos / KT, kb = pos - t * KT;
      const int ke = (kb + end - pos < KT) ? kb + end - pos : KT;
      u.tile = Tdp + t; u.kb = kb; u.ke = ke;
      u.mode = kb > 0 ? 1 : (ke < KT ? 2 : 0);
      u.slot = kb > 0 ? w : w + 1;
      pos += ke - kb;
    }
    return u;
  }
};

constexpr long long SK_PART_BYTES = 256 * 256 * 4;   // one parked 256x256 tile of fp32 accumulators
// scratch of a stream-K launch on `grid` workgroups: a parked tile per range boundary, then one 8-byte arrival flag per boundary
inline long long sk_ws_bytes(int grid) { return (long long)grid * SK_PART_BYTES + (long long)grid * 8; }

}  // namespace qamd
