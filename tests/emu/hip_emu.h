// TEST INFRASTRUCTURE ONLY -- never part of the product, never loaded by the package.
//
// A minimal SIMT emulator that lets the CPU test-suite execute the *same kernel source*
// (mi-gan_amd/csrc/migan_kernels.hpp) and the same host plan / C ABI (migan_host.hpp) without a GPU,
// to check tile indexing, LDS carving, barrier placement and the MFMA fragment mapping.
//
// Model: one workgroup = 64..1024 lanes (whole waves), each lane a user-level fiber (hand-rolled x86-64 context switch)
// on ONE OS thread; lanes run to the next collective (__syncthreads, MFMA, shuffle) and then yield
// round-robin.  Lane 0 therefore runs arbitrarily far ahead of lane 255 between barriers, which is
// the most adversarial legal schedule: a missing barrier shows up as a wrong result.  LDS is filled
// with NaNs before every block (reads of never-written LDS poison the output) and has a canary
// behind it.  Blocks are distributed over OS threads.
//
// MFMA semantics follow /opt/skills/guides/cdna_hip_programming.md section 3:
//   v_mfma_f32_32x32x2_f32: lane l supplies A[i=l&31][k=l>>5], B[k=l>>5][j=l&31];
//   D[i][j] = fma(A[i][1],B[1][j], fma(A[i][0],B[0][j], C[i][j])), lane holds column j=l&31,
//   rows i = (r&3) + 8*(r>>2) + 4*(l>>5) for r = 0..15.
#pragma once

#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

namespace hipemu {

struct Dim3 {
  unsigned x = 1, y = 1, z = 1;
};

// one LDS-DMA (buffer_load ... lds) of one lane: 16 bytes that land in LDS when the lane WAITS for them (MIGAN_WAIT_VMCNT), not when
// the instruction is issued -- a reader that is only ordered by a barrier, without the issuing wave's vmcnt wait, sees stale LDS
struct DmaOp {
  char* dst;
  unsigned char data[16];
  int bytes = 16;
};

struct Lane {
  Dim3 tid;
  void* sp = nullptr;
  bool done = false;
  std::vector<DmaOp> dma;      // outstanding LDS-DMAs, oldest first
};

constexpr int kMaxLanes = 1024;
constexpr int kMaxWaves = kMaxLanes / 64;
constexpr size_t kStackBytes = 96 * 1024;

struct Block {
  Dim3 bid, bdim, gdim;
  float* smem = nullptr;
  int nlanes = 0, cur = 0, alive = 0;
  Lane lanes[kMaxLanes];
  void* sched_sp = nullptr;
  int blk_arrived = 0;
  unsigned blk_gen = 0;
  int wave_arrived[kMaxWaves] = {0};
  unsigned wave_gen[kMaxWaves] = {0};
  int wave_alive[kMaxWaves] = {0};
  float xa[kMaxWaves][64], xb[kMaxWaves][64];
  unsigned xq[kMaxWaves][64][4], yq[kMaxWaves][64][4];
  float xf[kMaxWaves][64][8], yf[kMaxWaves][64][8];       // decoded 16-bit MFMA operands
  void (*invoke)(const void*) = nullptr;
  const void* arg = nullptr;
  char* stacks = nullptr;
};

extern thread_local Block* tl_blk;

extern "C" void hipemu_switch(void** save_sp, void* load_sp);

inline void switch_to(int from, int to) {
  Block* b = tl_blk;
  b->cur = to;
  hipemu_switch(&b->lanes[from].sp, b->lanes[to].sp);
}

// round-robin to the next live lane; returns immediately if alone
inline void yield_next() {
  Block* b = tl_blk;
  const int me = b->cur, n = b->nlanes;
  int nx = me;
  do {
    nx = (nx + 1 == n) ? 0 : nx + 1;
  } while (b->lanes[nx].done && nx != me);
  if (nx != me) switch_to(me, nx);
}

inline void block_barrier() {
  Block* b = tl_blk;
  const unsigned g = b->blk_gen;
  if (++b->blk_arrived >= b->alive) {
    b->blk_arrived = 0;
    b->blk_gen = g + 1;
  } else {
    while (b->blk_gen == g) yield_next();
  }
}

inline void wave_barrier() {
  Block* b = tl_blk;
  const int w = b->cur >> 6;
  const unsigned g = b->wave_gen[w];
  if (++b->wave_arrived[w] >= b->wave_alive[w]) {
    b->wave_arrived[w] = 0;
    b->wave_gen[w] = g + 1;
  } else {
    while (b->wave_gen[w] == g) yield_next();
  }
}

void run_grid(void (*invoke)(const void*), const void* arg, unsigned grid, unsigned block, size_t lds_bytes);

}  // namespace hipemu

// ---- the HIP surface the kernels use --------------------------------------------------------------
#define threadIdx (hipemu::tl_blk->lanes[hipemu::tl_blk->cur].tid)
#define blockIdx (hipemu::tl_blk->bid)
#define blockDim (hipemu::tl_blk->bdim)
#define gridDim (hipemu::tl_blk->gdim)

#define MIGAN_DEVICE
#define MIGAN_INLINE inline
#define MIGAN_GLOBAL
#define MIGAN_LAUNCH_BOUNDS(a, b)
#define MIGAN_DYN_SMEM(name) float* name = hipemu::tl_blk->smem
#define MIGAN_FMUL_RN(a, b) ((float)((a) * (b)))
#define MIGAN_FADD_RN(a, b) ((float)((a) + (b)))
#define MIGAN_FSUB_RN(a, b) ((float)((a) - (b)))
inline float __shfl_xor(float v, int mask);
// (lane-wise: the guarded code is a no-op in lanes where the predicate is false, so the wave-uniform branch of the product is only a speed-up)
#define MIGAN_ANY_LANE(p) (p)
#define MIGAN_COLD_PATH() do {} while (0)
#define MIGAN_CLAMP(v, lo, hi) fminf(fmaxf((v), (lo)), (hi))       // (like v_med3_f32: a NaN leaves it as lo; clamp4 / clamp1 repair it)
#define MIGAN_SWIZZLE_XOR(v, m) __shfl_xor((v), (m))
inline float hipemu_sum8(float v) {
  v += __shfl_xor(v, 1);
  v += __shfl_xor(v, 2);
  v += __shfl_xor(v, 4);
  return v;
}
#define MIGAN_SUM8(v) hipemu_sum8(v)
#define MIGAN_SCHED_FENCE() do {} while (0)
#define MIGAN_SCHED_GROUP(mask, n) do { (void)(mask); (void)(n); } while (0)
#define MIGAN_STORE_NT(ptr, v) (*(ptr) = (v))
#define MIGAN_LOAD_NT(ptr) (*(ptr))
#define MIGAN_OPAQUE(x) asm volatile("" : "+r"(x))
#define MIGAN_OPAQUE_F(x) asm volatile("" : "+x"(x))
#define MIGAN_CLOCK() 0
#define MIGAN_ATOMIC_ADD_U64(p, v) (*(p) += (v))

inline void __syncthreads() { hipemu::block_barrier(); }

inline float __shfl_xor(float v, int mask) {
  hipemu::Block* b = hipemu::tl_blk;
  const int w = b->cur >> 6, l = b->cur & 63;
  b->xa[w][l] = v;
  hipemu::wave_barrier();
  const float r = b->xa[w][l ^ mask];
  hipemu::wave_barrier();
  return r;
}

inline float hipemu_shfl(float v, int src) {
  hipemu::Block* b = hipemu::tl_blk;
  const int w = b->cur >> 6, l = b->cur & 63;
  b->xa[w][l] = v;
  hipemu::wave_barrier();
  const float r = b->xa[w][src & 63];
  hipemu::wave_barrier();
  return r;
}
#define MIGAN_READLANE(v, k) hipemu_shfl((v), (k))
#define MIGAN_OPAQUE_S(x) asm volatile("" : "+r"(x))

inline bool __all(bool pred) {
  hipemu::Block* b = hipemu::tl_blk;
  const int w = b->cur >> 6, l = b->cur & 63;
  b->xa[w][l] = pred ? 1.0f : 0.0f;
  hipemu::wave_barrier();
  bool r = true;
  for (int i = 0; i < 64; ++i)
    if (!b->lanes[(w << 6) + i].done && b->xa[w][i] == 0.0f) r = false;
  hipemu::wave_barrier();
  return r;
}

typedef float emu_f16v __attribute__((ext_vector_type(16)));
inline emu_f16v hipemu_mfma_f32_32x32x2(float a, float bv, emu_f16v c) {
  hipemu::Block* b = hipemu::tl_blk;
  const int w = b->cur >> 6, l = b->cur & 63;
  b->xa[w][l] = a;
  b->xb[w][l] = bv;
  hipemu::wave_barrier();
  const int j = l & 31, hh = l >> 5;
  for (int r = 0; r < 16; ++r) {
    const int i = (r & 3) + 8 * (r >> 2) + 4 * hh;
    c[r] = std::fmaf(b->xa[w][i + 32], b->xb[w][j + 32], std::fmaf(b->xa[w][i], b->xb[w][j], c[r]));
  }
  hipemu::wave_barrier();
  return c;
}
#define MIGAN_MFMA_F32_32X32X2(a, b, c) hipemu_mfma_f32_32x32x2((a), (b), (c))

// bf16 helpers and v_mfma_f32_32x32x16_bf16: lane l supplies A[i=l&31][k=8*(l>>5)+e] and
// B[k=8*(l>>5)+e][j=l&31], e = 0..7 (16 bytes per operand per lane); C/D layout as the f32 form.
inline unsigned hipemu_bf16_rne(float f) {
  unsigned u;
  std::memcpy(&u, &f, 4);
  return (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16;
}
#define MIGAN_PACK_BF16(lo, hi) (hipemu_bf16_rne(lo) | (hipemu_bf16_rne(hi) << 16))
typedef float emu_f4 __attribute__((ext_vector_type(4)));
// IEEE binary16 <-> binary32 in software (round to nearest even, subnormals, overflow to infinity): what
// v_cvt_pk_f16_f32 / v_cvt_f32_f16 do in the default rounding mode
inline unsigned hipemu_f16_rne(float f) {
  unsigned u;
  std::memcpy(&u, &f, 4);
  const unsigned sign = (u >> 16) & 0x8000u;
  const unsigned ex = (u >> 23) & 0xffu;
  unsigned man = u & 0x7fffffu;
  if (ex == 0xff) return sign | 0x7c00u | (man ? 0x200u : 0u);
  const int e = (int)ex - 127 + 15;
  if (e >= 31) return sign | 0x7c00u;
  if (e <= 0) {
    if (e < -10) return sign;                                  // below half the smallest subnormal
    man |= 0x800000u;
    const int sh = 14 - e;                                     // 14..24
    unsigned h = man >> sh;
    const unsigned rem = man & ((1u << sh) - 1), halfway = 1u << (sh - 1);
    if (rem > halfway || (rem == halfway && (h & 1))) ++h;
    return sign | h;
  }
  unsigned h = ((unsigned)e << 10) | (man >> 13);
  const unsigned rem = man & 0x1fffu;
  if (rem > 0x1000u || (rem == 0x1000u && (h & 1))) ++h;      // carry into the exponent is the right answer
  return sign | h;
}
inline float hipemu_f16_to_f32(unsigned h) {
  const unsigned sign = (h & 0x8000u) << 16, ex = (h >> 10) & 0x1fu, man = h & 0x3ffu;
  float f;
  if (ex == 0) {
    f = std::ldexp((float)man, -24);
  } else if (ex == 31) {
    const unsigned bits = 0x7f800000u | (man << 13);
    std::memcpy(&f, &bits, 4);
  } else {
    const unsigned bits = ((ex - 15 + 127) << 23) | (man << 13);
    std::memcpy(&f, &bits, 4);
  }
  return sign ? -f : f;
}
#define MIGAN_PACK_F16(lo, hi) (hipemu_f16_rne(lo) | (hipemu_f16_rne(hi) << 16))
#define MIGAN_F16LO_F32(pk) hipemu_f16_to_f32((pk) & 0xffffu)
#define MIGAN_F16HI_F32(pk) hipemu_f16_to_f32((pk) >> 16)
inline emu_f16v hipemu_mfma_16b_32x32x16(emu_f4 a, emu_f4 bv, emu_f16v c, bool f16) {
  hipemu::Block* b = hipemu::tl_blk;
  const int w = b->cur >> 6, l = b->cur & 63;
  // every lane decodes its own 8 + 8 operand elements once (into the exchange area, as floats), then reads the others'
  unsigned qa[4], qb[4];
  std::memcpy(qa, &a, 16);
  std::memcpy(qb, &bv, 16);
  auto elem = [f16](const unsigned* q, int e) {
    const unsigned pk = q[e >> 1];
    if (f16) return hipemu_f16_to_f32((e & 1) ? (pk >> 16) : (pk & 0xffffu));
    const unsigned bits = (e & 1) ? (pk & 0xffff0000u) : (pk << 16);
    float f;
    std::memcpy(&f, &bits, 4);
    return f;
  };
  float* fa = reinterpret_cast<float*>(b->xq[w][l]);      // 4 words of xq + 4 words of yq per lane are not enough for 16 floats:
  float* fb = reinterpret_cast<float*>(b->yq[w][l]);      // the decoded operands live in xf / yf below
  (void)fa; (void)fb;
  for (int e = 0; e < 8; ++e) {
    b->xf[w][l][e] = elem(qa, e);
    b->yf[w][l][e] = elem(qb, e);
  }
  hipemu::wave_barrier();
  const int j = l & 31, hh = l >> 5;
  for (int r = 0; r < 16; ++r) {
    const int i = (r & 3) + 8 * (r >> 2) + 4 * hh;
    double s = 0.0;
    for (int k = 0; k < 16; ++k)
      s += (double)b->xf[w][i + 32 * (k >> 3)][k & 7] * (double)b->yf[w][j + 32 * (k >> 3)][k & 7];
    c[r] = (float)((double)c[r] + s);
  }
  hipemu::wave_barrier();
  return c;
}
#define MIGAN_MFMA_BF16_32X32X16(a, b, c) hipemu_mfma_16b_32x32x16((a), (b), (c), false)
#define MIGAN_MFMA_F16_32X32X16(a, b, c) hipemu_mfma_16b_32x32x16((a), (b), (c), true)


// ---- LDS-DMA staging: buffer descriptor with the hardware's range check, and the direct global -> LDS copy.  The copy is DEFERRED:
// the bytes are captured at issue and written to LDS by MIGAN_WAIT_VMCNT(n) (all but the n newest of the lane's outstanding DMAs), so a
// missing or mis-counted wait leaves NaN-poisoned / stale LDS behind and fails the parity tests.  __syncthreads() does not drain them.
#define MIGAN_UNIFORM(x) (x)
struct MIGAN_BUF {
  const char* p;
  unsigned n;
};
#define MIGAN_MAKE_BUF(ptr, bytes) (MIGAN_BUF{reinterpret_cast<const char*>(ptr), (unsigned)(bytes)})
inline void hipemu_lds_dma16(MIGAN_BUF b, unsigned voff, unsigned soff, float* wave_base) {
  hipemu::Block* blk = hipemu::tl_blk;
  const int lane = blk->cur & 63;
  hipemu::DmaOp op;
  op.dst = reinterpret_cast<char*>(wave_base) + 16 * lane;
  if ((unsigned long long)voff + 16ull > (unsigned long long)b.n) std::memset(op.data, 0, 16);      // range check on the lane offset
  else std::memcpy(op.data, b.p + voff + soff, 16);
  blk->lanes[blk->cur].dma.push_back(op);
}
#define MIGAN_LDS_DMA16(buf, voff, soff, ldsp) hipemu_lds_dma16((buf), (voff), (soff), (ldsp))
inline void hipemu_lds_dma4(MIGAN_BUF b, unsigned voff, unsigned soff, float* wave_base) {       // buffer_load_dword ... lds
  hipemu::Block* blk = hipemu::tl_blk;
  const int lane = blk->cur & 63;
  hipemu::DmaOp op;
  op.bytes = 4;
  op.dst = reinterpret_cast<char*>(wave_base) + 4 * lane;
  if ((unsigned long long)voff + 4ull > (unsigned long long)b.n) std::memset(op.data, 0, 4);
  else std::memcpy(op.data, b.p + voff + soff, 4);
  blk->lanes[blk->cur].dma.push_back(op);
}
#define MIGAN_LDS_DMA4(buf, voff, soff, ldsp) hipemu_lds_dma4((buf), (voff), (soff), (ldsp))
// predicated forms: the hardware counts vector-memory operations per WAVE, this emulator per lane -- an inactive lane records an
// operation that writes nothing, so that a counted wait means the same on both
inline void hipemu_lds_dma_null() {
  hipemu::Block* blk = hipemu::tl_blk;
  hipemu::DmaOp op;
  op.bytes = 0;
  op.dst = nullptr;
  blk->lanes[blk->cur].dma.push_back(op);
}
#define MIGAN_LDS_DMA16_IF(cond, buf, voff, soff, ldsp) do { if (cond) hipemu_lds_dma16((buf), (voff), (soff), (ldsp)); else hipemu_lds_dma_null(); } while (0)
#define MIGAN_LDS_DMA4_IF(cond, buf, voff, soff, ldsp) do { if (cond) hipemu_lds_dma4((buf), (voff), (soff), (ldsp)); else hipemu_lds_dma_null(); } while (0)
inline void hipemu_wait_vmcnt(int n) {
  hipemu::Block* blk = hipemu::tl_blk;
  std::vector<hipemu::DmaOp>& q = blk->lanes[blk->cur].dma;
  const size_t keep = (size_t)(n < 0 ? 0 : n);
  if (q.size() <= keep) return;
  const size_t done = q.size() - keep;
  for (size_t i = 0; i < done; ++i)
    if (q[i].bytes) std::memcpy(q[i].dst, q[i].data, (size_t)q[i].bytes);
  q.erase(q.begin(), q.begin() + (long)done);
}
#define MIGAN_WAIT_VMCNT(n) hipemu_wait_vmcnt(n)
#define MIGAN_BARRIER_LDS() hipemu::block_barrier()
#define MIGAN_WAVE_SYNC() hipemu::wave_barrier()
#define MIGAN_SETPRIO(n) do {} while (0)

// ---- the runtime surface the host code uses ----------------------------------------------------------
namespace rt {
typedef void* stream_t;
struct event_t {
  std::chrono::steady_clock::time_point t;
  std::chrono::steady_clock::time_point* p = nullptr;
};

inline const char* backend_name() { return "emu:cpu-fibers (test only)"; }
inline std::string error_string(int rc) { return "emulator error " + std::to_string(rc); }
inline int set_device(int) { return 0; }
inline int get_device(int* dev) { *dev = 0; return 0; }
// the emulator executes every launch synchronously, in enqueue order: streams and ordering events are no-ops
inline int stream_create(stream_t* s) { *s = reinterpret_cast<stream_t>(0x1); return 0; }
inline int stream_destroy(stream_t) { return 0; }
inline int allow_dynamic_lds(const void*, size_t) { return 0; }

template <class Args>
struct Thunk {
  void (*kernel)(const Args);
  const Args* args;
  static void call(const void* self) {
    const Thunk* t = static_cast<const Thunk*>(self);
    t->kernel(*t->args);
  }
};

template <class Args>
inline int launch(void (*kernel)(const Args), const Args& a, unsigned grid, unsigned block, size_t lds, stream_t) {
  if (block == 0 || block % 64 != 0 || block > (unsigned)hipemu::kMaxLanes || lds > 160 * 1024) return 1;
  Thunk<Args> t{kernel, &a};
  hipemu::run_grid(&Thunk<Args>::call, &t, grid, block, lds);
  return 0;
}
inline int memcpy_d2h(void* dst, const void* src, size_t bytes, stream_t) {
  std::memcpy(dst, src, bytes);
  return 0;
}
inline int stream_sync(stream_t) { return 0; }
inline int event_create(event_t* e) {
  e->p = new std::chrono::steady_clock::time_point();
  return 0;
}
inline int event_create_sync(event_t* e) { e->p = nullptr; return 0; }
inline int event_destroy(event_t e) {
  delete e.p;
  return 0;
}
inline int stream_wait_event(stream_t, event_t) { return 0; }
inline int event_record(event_t e, stream_t) {
  if (e.p) *e.p = std::chrono::steady_clock::now();
  return 0;
}
inline int event_elapsed(float* ms, event_t a, event_t b) {
  *ms = std::chrono::duration<float, std::milli>(*b.p - *a.p).count();
  return 0;
}
}  // namespace rt
