// tests/hipemu/hipemu.cpp -- fiber scheduler behind hipemu.h (test infrastructure only).
#include "hipemu.h"

#include <execinfo.h>
#include <signal.h>
#include <unistd.h>
#include <vector>

namespace hipemu {

Fiber *cur = nullptr;
Dim3 g_blockIdx, g_blockDim, g_gridDim;
unsigned char *g_dyn_smem = nullptr;

namespace {

constexpr size_t kStack = 256 * 1024;
constexpr int kWave = 64;
constexpr size_t kSlot = 64;  // max bytes per lane per collective

struct WaveState {
    int alive = 0, arrived = 0;
    unsigned gen = 0;
    unsigned tag[2] = {0, 0};
    unsigned char buf[2][kWave * kSlot];
};

std::vector<Fiber> fibers;
std::vector<WaveState> waves;
int blk_alive = 0, blk_arrived = 0;
unsigned blk_gen = 0;
void *sched_sp = nullptr;
const std::function<void()> *body_fn = nullptr;

extern "C" void hipemu_switch(void **from_sp, void *to_sp);
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

void yield() { Fiber *f = cur; hipemu_switch(&f->sp, sched_sp); }

void trampoline()
{
    (*body_fn)();
    cur->done = true;
    yield();
    abort();  // a finished fiber is never resumed
}

void prepare(Fiber &f)
{
    uintptr_t top = ((uintptr_t)f.stack + kStack) & ~(uintptr_t)15;
    void **sp = (void **)top;
    *--sp = nullptr;                 // fake return address of trampoline
    *--sp = (void *)&trampoline;     // popped by hipemu_switch's ret
    for (int i = 0; i < 6; ++i) *--sp = nullptr;
    f.sp = sp;
}

[[noreturn]] void die(const char *msg)
{
    fprintf(stderr, "hipemu: %s (block %u thread %u)\n", msg, g_blockIdx.x, cur ? cur->tid.x : 0u);
    abort();
}

}  // namespace

[[noreturn]] void fail(const char *msg) { die(msg); }

void sync_block()
{
    unsigned gen = blk_gen;
    if (++blk_arrived == blk_alive) { blk_arrived = 0; ++blk_gen; return; }
    while (blk_gen == gen) yield();
}

static void wave_wait(WaveState &w)
{
    unsigned gen = w.gen;
    if (++w.arrived == w.alive) { w.arrived = 0; ++w.gen; return; }
    while (w.gen == gen) yield();
}

void wave_barrier_only()
{
    WaveState &w = waves[cur->wave];
    if (w.arrived == 0) w.tag[w.gen & 1] = 0xBA221E2u;
    else if (w.tag[w.gen & 1] != 0xBA221E2u) die("divergent wave collective (barrier vs other)");
    wave_wait(w);
}

const unsigned char *wave_gather(const void *in, size_t size, unsigned tag)
{
    if (size > kSlot) die("collective payload too large");
    WaveState &w = waves[cur->wave];
    unsigned par = w.gen & 1;
    if (w.arrived == 0) { w.tag[par] = tag; memset(w.buf[par], 0, sizeof(w.buf[par])); }
    else if (w.tag[par] != tag) die("divergent wave collective (different call sites)");
    memcpy(w.buf[par] + (size_t)cur->lane * size, in, size);
    wave_wait(w);
    return w.buf[par];
}

static void segv_handler(int sig)
{
    void *frames[48];
    int n = backtrace(frames, 48);
    fprintf(stderr, "hipemu: signal %d in block %u thread %u\n", sig, g_blockIdx.x, cur ? cur->tid.x : 0u);
    backtrace_symbols_fd(frames, n, 2);
    _exit(139);
}

void launch(Dim3 grid, Dim3 block, size_t smem, const std::function<void()> &body)
{
    static bool installed = false;
    if (!installed && getenv("HIPEMU_BACKTRACE")) { signal(SIGSEGV, segv_handler); installed = true; }
    if (cur) die("nested launch");
    unsigned nthreads = block.x * block.y * block.z;
    if (nthreads == 0 || nthreads > 1024) die("bad block size");
    if (fibers.size() < nthreads) {
        size_t old = fibers.size();
        fibers.resize(nthreads);
        for (size_t i = old; i < nthreads; ++i) fibers[i].stack = (char *)malloc(kStack);
    }
    std::vector<unsigned char> dyn(smem + 64);
    g_dyn_smem = (unsigned char *)(((uintptr_t)dyn.data() + 63) & ~(uintptr_t)63);
    g_blockDim = block;
    g_gridDim = grid;
    body_fn = &body;
    unsigned nwaves = (nthreads + kWave - 1) / kWave;
    for (unsigned bz = 0; bz < grid.z; ++bz)
    for (unsigned by = 0; by < grid.y; ++by)
    for (unsigned bx = 0; bx < grid.x; ++bx) {
        g_blockIdx = Dim3(bx, by, bz);
        waves.assign(nwaves, WaveState());
        blk_alive = (int)nthreads; blk_arrived = 0; blk_gen = 0;
        for (unsigned t = 0; t < nthreads; ++t) {
            Fiber &f = fibers[t];
            f.tid = Dim3(t % block.x, (t / block.x) % block.y, t / (block.x * block.y));
            f.lane = (int)(t % kWave);
            f.wave = (int)(t / kWave);
            f.done = false;
            waves[f.wave].alive++;
            prepare(f);
        }
        int remaining = (int)nthreads;
        while (remaining > 0) {
            int progressed = 0;
            for (unsigned t = 0; t < nthreads; ++t) {
                Fiber &f = fibers[t];
                if (f.done) continue;
                unsigned wg = waves[f.wave].gen, bg = blk_gen;
                int wa = waves[f.wave].arrived, ba = blk_arrived;
                cur = &f;
                hipemu_switch(&sched_sp, f.sp);
                cur = nullptr;
                if (f.done) {
                    --remaining; ++progressed;
                    WaveState &w = waves[f.wave];
                    --w.alive; --blk_alive;
                    if (w.arrived > 0) die("thread exited while its wave waits in a collective");
                    if (blk_arrived > 0 && blk_arrived == blk_alive) { blk_arrived = 0; ++blk_gen; }
                } else if (wg != waves[f.wave].gen || bg != blk_gen ||
                           wa != waves[f.wave].arrived || ba != blk_arrived) {
                    ++progressed;
                }
            }
            if (!progressed) die("deadlock: no fiber can make progress");
        }
    }
    body_fn = nullptr;
    g_dyn_smem = nullptr;
}

}  // namespace hipemu
