// tests/emu/emu_runtime.cpp -- TEST INFRASTRUCTURE ONLY: fiber scheduler behind hip/hip_runtime.h.
#include <hip/hip_runtime.h>
#include <ucontext.h>
#include <sys/mman.h>
#include <stdio.h>
#include <vector>
#include <mutex>

dim3 threadIdx, blockIdx, blockDim, gridDim;

namespace emu {

static const size_t STACK = 512 * 1024;

struct Fiber { ucontext_t ctx; char* stack; bool done; unsigned tid; };
struct WaveState { int arrive = 0; unsigned gen = 0; int alive = 0; uint64_t alive_mask = 0; uint64_t slots[64]; int row_arrive[4] = {0, 0, 0, 0}; unsigned row_gen[4] = {0, 0, 0, 0}; };

static ucontext_t g_sched;
static std::vector<Fiber> g_fibers;
static std::vector<WaveState> g_waves;
static int g_block_arrive = 0, g_block_alive = 0; static unsigned g_block_gen = 0;
static Fiber* g_cur = nullptr;
static const std::function<void()>* g_body = nullptr;
static std::vector<char> g_dyn;
static std::vector<char*> g_stack_pool;

void* dyn_smem() { return g_dyn.data(); }
uint64_t* wave_slots() { return g_waves[g_cur->tid >> 6].slots; }
uint64_t wave_alive_mask() { return g_waves[g_cur->tid >> 6].alive_mask; }

// AddressSanitizer must be told about every stack switch (it tracks the bounds of the running stack)
#if defined(__SANITIZE_ADDRESS__)
extern "C" void __sanitizer_start_switch_fiber(void** fake_stack_save, const void* bottom, size_t size);
extern "C" void __sanitizer_finish_switch_fiber(void* fake_stack_save, const void** bottom_old, size_t* size_old);
static const void* g_sched_stack = nullptr; static size_t g_sched_size = 0;
static void to_sched(Fiber* f, bool dying) { void* fake = nullptr; __sanitizer_start_switch_fiber(dying ? nullptr : &fake, g_sched_stack, g_sched_size); swapcontext(&f->ctx, &g_sched); __sanitizer_finish_switch_fiber(fake, nullptr, nullptr); }
static void to_fiber(Fiber* f) { void* fake = nullptr; __sanitizer_start_switch_fiber(&fake, f->stack, STACK); swapcontext(&g_sched, &f->ctx); __sanitizer_finish_switch_fiber(fake, nullptr, nullptr); }
static void fiber_entered() { __sanitizer_finish_switch_fiber(nullptr, &g_sched_stack, &g_sched_size); }
#else
static void to_sched(Fiber* f, bool) { swapcontext(&f->ctx, &g_sched); }
static void to_fiber(Fiber* f) { swapcontext(&g_sched, &f->ctx); }
static void fiber_entered() {}
#endif
static void yield() { Fiber* f = g_cur; to_sched(f, false); threadIdx.x = f->tid; }

void sync_wave()
{
    WaveState& w = g_waves[g_cur->tid >> 6];
    unsigned my = w.gen;
    if (++w.arrive == w.alive) { w.arrive = 0; ++w.gen; return; }
    while (w.gen == my) yield();
}

static int row_alive(const WaveState& w, int row) { return __builtin_popcountll(w.alive_mask >> (16 * row) & 0xffffull); }

void sync_row()
{
    WaveState& w = g_waves[g_cur->tid >> 6];
    const int row = (g_cur->tid & 63) >> 4;
    unsigned my = w.row_gen[row];
    if (++w.row_arrive[row] == row_alive(w, row)) { w.row_arrive[row] = 0; ++w.row_gen[row]; return; }
    while (w.row_gen[row] == my) yield();
}

void sync_block()
{
    unsigned my = g_block_gen;
    if (++g_block_arrive == g_block_alive) { g_block_arrive = 0; ++g_block_gen; return; }
    while (g_block_gen == my) yield();
}

static void trampoline()
{
    fiber_entered();
    (*g_body)();
    Fiber* f = g_cur;
    f->done = true;
    WaveState& w = g_waves[f->tid >> 6];
    --w.alive; w.alive_mask &= ~(1ull << (f->tid & 63));
    if (w.alive > 0 && w.arrive == w.alive) { w.arrive = 0; ++w.gen; }
    { const int row = (f->tid & 63) >> 4; const int ra = row_alive(w, row); if (ra > 0 && w.row_arrive[row] == ra) { w.row_arrive[row] = 0; ++w.row_gen[row]; } }
    --g_block_alive;
    if (g_block_alive > 0 && g_block_arrive == g_block_alive) { g_block_arrive = 0; ++g_block_gen; }
    to_sched(f, true);
}

static std::mutex g_launch_mu;     // the fiber scheduler state is process-global: launches from several host threads are serialised

void launch(dim3 grid, dim3 block, size_t shmem, const std::function<void()>& body)
{
    std::lock_guard<std::mutex> lk(g_launch_mu);
    const unsigned nt = block.x;
    gridDim = grid; blockDim = block;
    g_body = &body;
    while (g_stack_pool.size() < nt) {
        char* s = (char*)mmap(0, STACK, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
        g_stack_pool.push_back(s);
    }
    g_fibers.resize(nt);
    for (unsigned b = 0; b < grid.x; ++b) {
        blockIdx = dim3(b);
        g_dyn.assign(shmem + 64, (char)0xCD);
        g_waves.assign((nt + 63) / 64, WaveState());
        g_block_arrive = 0; g_block_alive = (int)nt; g_block_gen = 0;
        for (unsigned t = 0; t < nt; ++t) {
            Fiber& f = g_fibers[t];
            f.stack = g_stack_pool[t]; f.done = false; f.tid = t;
            getcontext(&f.ctx);
            f.ctx.uc_stack.ss_sp = f.stack; f.ctx.uc_stack.ss_size = STACK; f.ctx.uc_link = 0;
            makecontext(&f.ctx, (void (*)())trampoline, 0);
            WaveState& w = g_waves[t >> 6];
            ++w.alive; w.alive_mask |= 1ull << (t & 63);
        }
        unsigned remaining = nt;
        while (remaining) {
            unsigned progressed = 0;
            for (unsigned t = 0; t < nt; ++t) {
                Fiber& f = g_fibers[t];
                if (f.done) continue;
                g_cur = &f; threadIdx = dim3(t);
                to_fiber(&f);
                if (f.done) { --remaining; }
                ++progressed;
            }
            if (!progressed) break;
        }
    }
    g_body = nullptr;
}

} // namespace emu

// k_index.hip (the device half of the index builder, rocPRIM sorts) is not part of the emulation build: the host builder runs
#include <string>
#include <vector>
#include "index_io.h"
bool device_index_available() { return false; }
bool device_index_pieces(const std::vector<uint8_t>&, IndexPieces&, std::string* err) { if (err) *err = "no device in the emulation build"; return false; }
