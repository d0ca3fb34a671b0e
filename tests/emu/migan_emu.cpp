// TEST INFRASTRUCTURE ONLY: builds libmigan_emu.so = the product's kernel source + host plan + C ABI
// compiled for the CPU on top of the fiber SIMT emulator in hip_emu.h.  Used by tests/test_emu_*.py.
#include "hip_emu.h"

#include "../../mi-gan_amd/csrc/migan_kernels.hpp"
#include "../../mi-gan_amd/csrc/comodgan_kernels.hpp"
#include "../../mi-gan_amd/csrc/migan_table.hpp"
#include "../../mi-gan_amd/csrc/migan_pipe.hpp"
#include "../../mi-gan_amd/csrc/migan_pipe_table.inc"
#include "../../mi-gan_amd/csrc/migan_wide2.hpp"
#include "../../mi-gan_amd/csrc/migan_wide2_table.inc"
// every slice of the sepconv_kernel table (the product compiles one translation unit per slice)
#define MIGAN_SLICE_G 0
#define MIGAN_SLICE_S 0
#include "../../mi-gan_amd/csrc/migan_k_slice.inc"
#undef MIGAN_SLICE_G
#define MIGAN_SLICE_G 1
#include "../../mi-gan_amd/csrc/migan_k_slice.inc"
#undef MIGAN_SLICE_G
#define MIGAN_SLICE_G 2
#include "../../mi-gan_amd/csrc/migan_k_slice.inc"
#undef MIGAN_SLICE_S
#define MIGAN_SLICE_S 1
#include "../../mi-gan_amd/csrc/migan_k_slice.inc"
#undef MIGAN_SLICE_S
#define MIGAN_SLICE_S 2
#include "../../mi-gan_amd/csrc/migan_k_slice.inc"
#undef MIGAN_SLICE_G
#define MIGAN_SLICE_G 3
#include "../../mi-gan_amd/csrc/migan_k_slice.inc"
#undef MIGAN_SLICE_S
#define MIGAN_SLICE_S 1
#include "../../mi-gan_amd/csrc/migan_k_slice.inc"
#undef MIGAN_SLICE_S
#undef MIGAN_SLICE_G
#include "../../mi-gan_amd/csrc/migan_host.hpp"
#include "../../mi-gan_amd/csrc/comodgan_host.hpp"

#include <mutex>

asm(R"(
.text
.globl hipemu_switch
.type hipemu_switch,@function
hipemu_switch:
    pushq %rbp
    pushq %rbx
    pushq %r12
    pushq %r13
    pushq %r14
    pushq %r15
    movq %rsp, (%rdi)
    movq %rsi, %rsp
    popq %r15
    popq %r14
    popq %r13
    popq %r12
    popq %rbx
    popq %rbp
    ret
.size hipemu_switch,.-hipemu_switch
)");

namespace hipemu {

thread_local Block* tl_blk = nullptr;

static void lane_exit() {
  Block* b = tl_blk;
  const int me = b->cur, w = me >> 6;
  b->lanes[me].done = true;
  b->alive--;
  b->wave_alive[w]--;
  // a lane that left must not strand the others at a barrier
  if (b->alive > 0 && b->blk_arrived > 0 && b->blk_arrived >= b->alive) {
    b->blk_arrived = 0;
    b->blk_gen++;
  }
  if (b->wave_alive[w] > 0 && b->wave_arrived[w] > 0 && b->wave_arrived[w] >= b->wave_alive[w]) {
    b->wave_arrived[w] = 0;
    b->wave_gen[w]++;
  }
  void* dummy;
  for (int k = 1; k <= b->nlanes; ++k) {
    const int nx = (me + k) % b->nlanes;
    if (!b->lanes[nx].done) {
      b->cur = nx;
      hipemu_switch(&dummy, b->lanes[nx].sp);
    }
  }
  hipemu_switch(&dummy, b->sched_sp);
  std::abort();
}

static void lane_entry() {
  Block* b = tl_blk;
  b->invoke(b->arg);
  lane_exit();
}

struct Worker {
  Block* blk = nullptr;
  char* stacks = nullptr;
  float* smem = nullptr;
};

static std::vector<Worker>& workers() {
  static std::vector<Worker> w;
  return w;
}
static std::mutex g_launch_mutex;

static void run_block(Worker& wk, void (*invoke)(const void*), const void* arg, unsigned bid, unsigned grid, unsigned block,
                      size_t lds_bytes) {
  Block* b = wk.blk;
  b->bid = Dim3{bid, 0, 0};
  b->bdim = Dim3{block, 1, 1};
  b->gdim = Dim3{grid, 1, 1};
  b->nlanes = (int)block;
  b->alive = (int)block;
  b->cur = 0;
  b->blk_arrived = 0;
  b->blk_gen = 0;
  b->invoke = invoke;
  b->arg = arg;
  const size_t nf = lds_bytes / sizeof(float);
  const float qnan = std::nanf("");
  for (size_t i = 0; i < nf; ++i) wk.smem[i] = qnan;
  for (size_t i = nf; i < nf + 64; ++i) wk.smem[i] = 12345.0f;   // canary
  b->smem = wk.smem;
  for (int w = 0; w < kMaxWaves; ++w) {
    b->wave_arrived[w] = 0;
    b->wave_gen[w] = 0;
    b->wave_alive[w] = 0;
  }
  for (int i = 0; i < (int)block; ++i) {
    Lane& l = b->lanes[i];
    l.tid = Dim3{(unsigned)i, 0, 0};
    l.done = false;
    l.dma.clear();
    b->wave_alive[i >> 6]++;
    uintptr_t top = reinterpret_cast<uintptr_t>(wk.stacks + (size_t)(i + 1) * kStackBytes) & ~(uintptr_t)15;
    void** slot = reinterpret_cast<void**>(top - 16);   // return address: entry sees rsp % 16 == 8
    *slot = reinterpret_cast<void*>(&lane_entry);
    void** sp = slot - 6;                                // r15 r14 r13 r12 rbx rbp
    for (int k = 0; k < 6; ++k) sp[k] = nullptr;
    l.sp = sp;
  }
  tl_blk = b;
  hipemu_switch(&b->sched_sp, b->lanes[0].sp);
  for (size_t i = nf; i < nf + 64; ++i)
    if (wk.smem[i] != 12345.0f) {
      std::fprintf(stderr, "hipemu: LDS overrun in block %u (float index %zu past %zu)\n", bid, i, nf);
      std::abort();
    }
}

void run_grid(void (*invoke)(const void*), const void* arg, unsigned grid, unsigned block, size_t lds_bytes) {
  std::lock_guard<std::mutex> lock(g_launch_mutex);
  unsigned nthreads = std::thread::hardware_concurrency();
  if (const char* e = std::getenv("MIGAN_EMU_THREADS")) nthreads = (unsigned)std::atoi(e);
  if (nthreads < 1) nthreads = 1;
  if (nthreads > grid) nthreads = grid;
  auto& ws = workers();
  if (ws.size() < nthreads) ws.resize(nthreads);
  const size_t need_floats = 160 * 1024 / sizeof(float) + 64;
  for (unsigned i = 0; i < nthreads; ++i) {
    if (!ws[i].blk) {
      ws[i].blk = new Block();
      ws[i].stacks = static_cast<char*>(std::malloc((size_t)kMaxLanes * kStackBytes + 64));
      ws[i].smem = static_cast<float*>(std::aligned_alloc(64, need_floats * sizeof(float)));
    }
  }
  std::atomic<unsigned> next{0};
  auto body = [&](unsigned wi) {
    for (;;) {
      const unsigned bid = next.fetch_add(1);
      if (bid >= grid) break;
      run_block(ws[wi], invoke, arg, bid, grid, block, lds_bytes);
    }
  };
  if (nthreads == 1) {
    body(0);
  } else {
    std::vector<std::thread> th;
    for (unsigned i = 0; i < nthreads; ++i) th.emplace_back(body, i);
    for (auto& t : th) t.join();
  }
}

}  // namespace hipemu
