// hip_sim.cpp -- TEST INFRASTRUCTURE ONLY: the fibre scheduler behind tests/sim/hip_sim.h (see there).
#include "hip_sim.h"
#include <execinfo.h>
#include <signal.h>
#include <unistd.h>

KsSimIdx threadIdx, blockIdx, blockDim, gridDim;

namespace ks_sim {
namespace {
extern "C" void ks_sim_switch(void** save_sp, void* load_sp);
asm(R"(
.text
.globl ks_sim_switch
.type ks_sim_switch,@function
ks_sim_switch:
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
.size ks_sim_switch,.-ks_sim_switch
)");

constexpr size_t kStack = 512 * 1024;
constexpr unsigned kMaxThreads = 1024;
enum Wait { RUN = 0, WAVE = 1, BLOCK = 2, DONE = 3, YIELDED = 4 };
struct Wave { uint64_t vals[2][64]; uint64_t present[2]; unsigned arrived = 0, gen = 0, nlive = 0; };
struct Fiber { void* sp = nullptr; Wait wait = RUN; unsigned waitgen = 0; unsigned tid = 0; unsigned wgen = 0; void* site = nullptr; };
struct Block {
  std::vector<Fiber> f; std::vector<Wave> w; unsigned nthreads = 0, b_arrived = 0, b_gen = 0, b_nlive = 0;
  const std::function<void()>* body = nullptr; void* sched_sp = nullptr; unsigned cur = 0, pass = 0;
};
Block* g_blk = nullptr;
char* g_stacks = nullptr;
std::mutex g_mu;

void yield_to_sched() { Fiber& me = g_blk->f[g_blk->cur]; ks_sim_switch(&me.sp, g_blk->sched_sp); }
void leave_wave(Block& B, unsigned tid) {
  Wave& W = B.w[tid >> 6];
  --W.nlive;
  if (W.nlive && W.arrived == W.nlive) { W.arrived = 0; ++W.gen; }        // the others were waiting for this lane only
  --B.b_nlive;
  if (B.b_nlive && B.b_arrived == B.b_nlive) { B.b_arrived = 0; ++B.b_gen; }
}
extern "C" void ks_sim_trampoline() {
  Block& B = *g_blk; const unsigned tid = B.cur;
  (*B.body)();
  B.f[tid].wait = DONE;
  leave_wave(B, tid);
  yield_to_sched();
  abort();
}
bool runnable(const Block& B, const Fiber& f) {
  if (f.wait == RUN) return true;
  if (f.wait == WAVE) return B.w[f.tid >> 6].gen != f.waitgen;
  if (f.wait == BLOCK) return B.b_gen != f.waitgen;
  if (f.wait == YIELDED) return B.pass != f.waitgen;      // (a polling loop's turn is over until the scheduler has gone round the other waves)
  return false;
}
}  // namespace

int lane_id() { return (int)(g_blk->cur & 63u); }

const uint64_t* exchange(uint64_t v, uint64_t* present) {
  Block& B = *g_blk; Fiber& me = B.f[B.cur]; Wave& W = B.w[me.tid >> 6];
  me.site = __builtin_return_address(0);
  const unsigned par = me.wgen & 1u, lane = me.tid & 63u;
  if (W.arrived == 0) W.present[par] = 0;
  W.vals[par][lane] = v; W.present[par] |= 1ull << lane;
  const unsigned g = W.gen;
  if (++W.arrived == W.nlive) { W.arrived = 0; ++W.gen; }
  else { me.wait = WAVE; me.waitgen = g; yield_to_sched(); me.wait = RUN; }
  ++me.wgen;
  *present = W.present[par];
  return W.vals[par];
}

// A lane in a polling loop (waiting for another wave to write something, no barrier in between) gives the other waves a turn: on the GPU that is an s_sleep.
void yield() { Block& B = *g_blk; Fiber& me = B.f[B.cur]; me.site = __builtin_return_address(0); me.wait = YIELDED; me.waitgen = B.pass; yield_to_sched(); me.wait = RUN; }

void block_barrier() {
  Block& B = *g_blk; Fiber& me = B.f[B.cur];
  me.site = __builtin_return_address(0);
  const unsigned g = B.b_gen;
  if (++B.b_arrived == B.b_nlive) { B.b_arrived = 0; ++B.b_gen; }
  else { me.wait = BLOCK; me.waitgen = g; yield_to_sched(); me.wait = RUN; }
}

void launch(dim3 grid, dim3 block, const std::function<void()>& body) {
  std::lock_guard<std::mutex> lock(g_mu);
  const unsigned nthreads = block.x * block.y * block.z;
  if (nthreads == 0 || nthreads > kMaxThreads) { fprintf(stderr, "ks_sim: bad block size %u\n", nthreads); abort(); }
  if (!g_stacks) {
    g_stacks = (char*)mmap(nullptr, kStack * kMaxThreads, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
    if (g_stacks == (char*)MAP_FAILED) { perror("ks_sim: mmap"); abort(); }
  }
  // KS_SIM_ALARM=<seconds>: a kernel that spins (no collective, no barrier: the deadlock report never fires) is stopped and says where the running lane stands
  if (const char* al = getenv("KS_SIM_ALARM")) {
    signal(SIGALRM, [](int) { void* bt[48]; const int n = backtrace(bt, 48); const char m[] = "ks_sim: KS_SIM_ALARM went off; the running lane stands at (addr2line -e <libksolve.so> <address minus the load address>):\n";
                              (void)!write(2, m, sizeof m - 1); backtrace_symbols_fd(bt, n, 2); _exit(3); });
    alarm((unsigned)atoi(al));
  }
  static unsigned seed = getenv("KS_SIM_SEED") ? (unsigned)atoi(getenv("KS_SIM_SEED")) : 0u;
  gridDim = {grid.x, grid.y, grid.z}; blockDim = {block.x, block.y, block.z};
  for (unsigned bz = 0; bz < grid.z; ++bz) for (unsigned by = 0; by < grid.y; ++by) for (unsigned bx = 0; bx < grid.x; ++bx) {
    Block B; B.nthreads = nthreads; B.body = &body; B.f.resize(nthreads); B.w.resize((nthreads + 63) / 64); B.b_nlive = nthreads;
    for (unsigned t = 0; t < nthreads; ++t) {
      Fiber& f = B.f[t]; f.tid = t; B.w[t >> 6].nlive++;
      char* top = g_stacks + (size_t)(t + 1) * kStack;
      void** sp = (void**)(top - 64);
      for (int i = 0; i < 6; ++i) sp[i] = nullptr;
      sp[6] = (void*)&ks_sim_trampoline; sp[7] = nullptr;
      f.sp = sp;
    }
    g_blk = &B;
    const unsigned nw = (unsigned)B.w.size();
    unsigned done = 0, wave0 = seed % nw;
    while (done < nthreads) {
      bool any = false; ++B.pass;
      for (unsigned wi = 0; wi < nw; ++wi) {
        const unsigned w = (wave0 + wi) % nw;
        bool progressed = true;
        while (progressed) {
          progressed = false;
          const unsigned lo = w * 64, hi = std::min(nthreads, lo + 64);
          for (unsigned t = lo; t < hi; ++t) {
            Fiber& f = B.f[t];
            if (!runnable(B, f)) continue;
            B.cur = t; blockIdx = {bx, by, bz}; threadIdx = {t % block.x, (t / block.x) % block.y, t / (block.x * block.y)};
            ks_sim_switch(&B.sched_sp, f.sp);
            progressed = true; any = true;
            if (f.wait == DONE) ++done;
          }
        }
      }
      if (seed) { seed = seed * 1664525u + 1013904223u; wave0 = (seed >> 16) % nw; }
      if (!any && done < nthreads) {
        fprintf(stderr, "ks_sim: deadlock in block (%u,%u,%u): %u of %u threads finished; a collective was reached in divergent control flow, or a barrier is missing a wave\n", bx, by, bz, done, nthreads);
        for (unsigned t = 0; t < nthreads; ++t) if (B.f[t].wait != DONE) { fprintf(stderr, "  first stuck thread %u: wait kind %d\n", t, (int)B.f[t].wait); break; }
        for (unsigned w = 0; w < nw; ++w) { unsigned c[5] = {0, 0, 0, 0, 0}; for (unsigned t = w * 64; t < std::min(nthreads, w * 64 + 64); ++t) c[(int)B.f[t].wait]++; fprintf(stderr, "  wave %u: %u runnable, %u at a wave collective, %u at the block barrier, %u done\n", w, c[0], c[1], c[2], c[3]);
          if (c[1] && c[2]) { std::map<void*, unsigned> sites; for (unsigned t = w * 64; t < std::min(nthreads, w * 64 + 64); ++t) sites[B.f[t].site]++; for (auto& kv : sites) { if (kv.second <= 4) for (unsigned t = w * 64; t < std::min(nthreads, w * 64 + 64); ++t) if (B.f[t].site == kv.first) fprintf(stderr, "    (lane %u)\n", t & 63); } for (auto& kv : sites) fprintf(stderr, "    %u lanes wait at call site %p (addr2line -e <libksolve.so> <that minus the library's load address; /proc/self/maps below>)\n", kv.second, kv.first); } }
        { FILE* mf = fopen("/proc/self/maps", "r"); char ln[512]; while (mf && fgets(ln, sizeof ln, mf)) if (strstr(ln, "libksolve") && strstr(ln, "r-xp")) fputs(ln, stderr); if (mf) fclose(mf); }
        abort();
      }
    }
    g_blk = nullptr;
  }
}
}  // namespace ks_sim
