/*
 * tests/emu/emu.cpp -- coroutine scheduler behind tests/emu/hip/hip_runtime.h
 * (TEST INFRASTRUCTURE ONLY; see that header).
 */
#include "hip/hip_runtime.h"

#include <ucontext.h>

/* Context switches are what this harness does (64 per cross-lane operation, 512-1 024 lanes in a workgroup-per-chunk launch),
 * and glibc's swapcontext saves and restores the signal mask with two system calls a switch: a third of the CPU test tier
 * was spent in the kernel. On x86-64 (and outside sanitizer builds, which want to see ucontext) the switch is six pushes,
 * a stack swap and six pops. */
#if defined(__x86_64__) && !defined(__SANITIZE_ADDRESS__) && !defined(EMU_USE_UCONTEXT)
#define EMU_FAST_SWITCH 1
extern "C" void emu_switch(void** save_sp, void* load_sp);
asm(R"(
.text
.globl emu_switch
.type emu_switch,@function
emu_switch:
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
.size emu_switch,.-emu_switch
)");
#else
#define EMU_FAST_SWITCH 0
#endif

#include <memory>
#include <vector>

namespace emu {

namespace {

constexpr size_t kStackBytes = 256 * 1024;

enum class State { Ready, WaitWave, WaitBlock, Done };

struct LaneRt
{
  Lane pub;
#if EMU_FAST_SWITCH
  void* sp = nullptr;
#else
  ucontext_t ctx;
#endif
  std::unique_ptr<char[]> stack;
  State state = State::Ready;
  int wait_op = 0;
};

struct WaveRt
{
  Contribution contrib[2][64];
  int parity = 0;          /* buffer the next rendezvous writes into */
  uint64_t live = 0;       /* lanes not yet finished */
  uint64_t arrived = 0;
  uint64_t last_live = 0;  /* participants of the last completed rendezvous */
  int op_id = 0;
};

struct BlockRt
{
  std::vector<LaneRt> lanes;
  std::vector<WaveRt> waves;
  size_t block_arrived = 0;
  size_t block_live = 0;
};

#if EMU_FAST_SWITCH
void* g_sched_sp = nullptr;
#else
ucontext_t g_sched_ctx;
#endif
BlockRt* g_block = nullptr;
LaneRt* g_cur = nullptr;
dim3 g_bdim, g_gdim;
const std::function<void()>* g_body = nullptr;
uint64_t g_rng = 0x9E3779B97F4A7C15ull;

uint64_t next_rand()
{
  g_rng ^= g_rng << 13;
  g_rng ^= g_rng >> 7;
  g_rng ^= g_rng << 17;
  return g_rng;
}

inline void to_scheduler(LaneRt* me)
{
#if EMU_FAST_SWITCH
  emu_switch(&me->sp, g_sched_sp);
#else
  swapcontext(&me->ctx, &g_sched_ctx);
#endif
}

void lane_entry()
{
  (*g_body)();
  g_cur->state = State::Done;
  to_scheduler(g_cur);
}

[[noreturn]] void die(const char* msg)
{
  fprintf(stderr, "emu: %s\n", msg);
  abort();
}

void complete_wave(WaveRt& w, BlockRt& b, int wave_index)
{
  w.last_live = w.arrived;
  w.arrived = 0;
  w.parity ^= 1;
  for (auto& l : b.lanes) {
    if (l.pub.wave == wave_index && l.state == State::WaitWave) {
      l.state = State::Ready;
    }
  }
}

} // namespace

alignas(16) static uint8_t g_dynamic_lds[160 * 1024];
uint8_t* dynamic_lds() { return g_dynamic_lds; }
Lane* cur() { return &g_cur->pub; }
const dim3& block_dim() { return g_bdim; }
const dim3& grid_dim() { return g_gdim; }
void set_seed(uint64_t seed) { g_rng = seed * 0x9E3779B97F4A7C15ull + 1; }

void wave_rendezvous(int op_id, uint64_t a, uint64_t b)
{
  LaneRt* me = g_cur;
  WaveRt& w = g_block->waves[me->pub.wave];
  if (w.arrived == 0) {
    w.op_id = op_id;
  } else if (w.op_id != op_id) {
    fprintf(stderr, "emu: lanes of one wave met at different cross-lane ops (%d vs %d)\n", w.op_id, op_id);
    abort();
  }
  w.contrib[w.parity][me->pub.lane] = Contribution{a, b};
  w.arrived |= 1ull << me->pub.lane;
  me->state = State::WaitWave;
  me->wait_op = op_id;
  if (w.arrived == w.live) {
    complete_wave(w, *g_block, me->pub.wave);
  }
  to_scheduler(me);
}

const Contribution& peer(int lane)
{
  WaveRt& w = g_block->waves[g_cur->pub.wave];
  return w.contrib[w.parity ^ 1][lane & 63];
}

uint64_t live_mask()
{
  return g_block->waves[g_cur->pub.wave].last_live;
}

void block_rendezvous()
{
  LaneRt* me = g_cur;
  BlockRt& b = *g_block;
  me->state = State::WaitBlock;
  b.block_arrived++;
  if (b.block_arrived == b.block_live) {
    b.block_arrived = 0;
    for (auto& l : b.lanes) {
      if (l.state == State::WaitBlock) {
        l.state = State::Ready;
      }
    }
  }
  to_scheduler(me);
}

void launch(const std::function<void()>& body, dim3 grid, dim3 block)
{
  if (g_block != nullptr) {
    die("nested launch");
  }
  const size_t nthreads = (size_t)block.x * block.y * block.z;
  if (nthreads == 0 || nthreads > 1024) {
    die("bad block size");
  }
  g_bdim = block;
  g_gdim = grid;
  g_body = &body;
  const size_t nwaves = (nthreads + 63) / 64;
  BlockRt b;
  b.lanes.resize(nthreads);
  b.waves.resize(nwaves);
  for (auto& l : b.lanes) {
    l.stack.reset(new char[kStackBytes]);
  }
  std::vector<size_t> order(nthreads);
  for (unsigned bz = 0; bz < grid.z; ++bz) {
    for (unsigned by = 0; by < grid.y; ++by) {
      for (unsigned bx = 0; bx < grid.x; ++bx) {
        for (auto& w : b.waves) {
          w.live = 0;
          w.arrived = 0;
          w.parity = 0;
          w.last_live = 0;
        }
        b.block_arrived = 0;
        b.block_live = nthreads;
        for (size_t t = 0; t < nthreads; ++t) {
          LaneRt& l = b.lanes[t];
          l.pub.tid = Idx3{(unsigned)(t % block.x), (unsigned)((t / block.x) % block.y), (unsigned)(t / ((size_t)block.x * block.y))};
          l.pub.bid = Idx3{bx, by, bz};
          l.pub.lane = (int)(t & 63);
          l.pub.wave = (int)(t >> 6);
          l.state = State::Ready;
          b.waves[t >> 6].live |= 1ull << (t & 63);
#if EMU_FAST_SWITCH
          {
            /* a fresh stack: six zeroed callee-saved registers, then lane_entry as the address emu_switch returns to (the
             * stack pointer is 8 modulo 16 there, as after a call) */
            uintptr_t top = ((uintptr_t)l.stack.get() + kStackBytes) & ~(uintptr_t)15;
            void** spp = (void**)(top - 8);
            *--spp = (void*)&lane_entry;
            for (int r = 0; r < 6; ++r) {
              *--spp = nullptr;
            }
            l.sp = (void*)spp;
          }
#else
          getcontext(&l.ctx);
          l.ctx.uc_stack.ss_sp = l.stack.get();
          l.ctx.uc_stack.ss_size = kStackBytes;
          l.ctx.uc_link = nullptr;
          makecontext(&l.ctx, (void (*)())lane_entry, 0);
#endif
        }
        g_block = &b;
        size_t done = 0;
        while (done < nthreads) {
          /* one sweep over the lanes in a fresh pseudo-random order */
          for (size_t t = 0; t < nthreads; ++t) {
            order[t] = t;
          }
          for (size_t t = nthreads; t > 1; --t) {
            std::swap(order[t - 1], order[next_rand() % t]);
          }
          bool progressed = false;
          for (size_t oi = 0; oi < nthreads; ++oi) {
            LaneRt& l = b.lanes[order[oi]];
            if (l.state != State::Ready) {
              continue;
            }
            progressed = true;
            g_cur = &l;
#if EMU_FAST_SWITCH
            emu_switch(&g_sched_sp, l.sp);
#else
            swapcontext(&g_sched_ctx, &l.ctx);
#endif
            g_cur = nullptr;
            if (l.state == State::Done) {
              ++done;
              WaveRt& w = b.waves[l.pub.wave];
              w.live &= ~(1ull << l.pub.lane);
              if (w.live != 0 && w.arrived == w.live) {
                complete_wave(w, b, l.pub.wave);
              }
              b.block_live--;
              if (b.block_live != 0 && b.block_arrived == b.block_live) {
                b.block_arrived = 0;
                for (auto& o : b.lanes) {
                  if (o.state == State::WaitBlock) {
                    o.state = State::Ready;
                  }
                }
              }
            }
          }
          if (!progressed) {
            die("deadlock: lanes are waiting at cross-lane operations that not all live lanes reach");
          }
        }
        g_block = nullptr;
      }
    }
  }
  g_body = nullptr;
}

} // namespace emu

/* ---- statistics hook --------------------------------------------------------- */
#include <map>
#include <string>
static std::map<std::string, unsigned long long> g_stats;
extern "C" void emu_stat_add(const char* name, unsigned long long n) { g_stats[name] += n; }
extern "C" void emu_stats_dump(int reset)
{
  for (auto& kv : g_stats) {
    fprintf(stderr, "stat %-28s %llu\n", kv.first.c_str(), kv.second);
  }
  if (reset) {
    g_stats.clear();
  }
}
