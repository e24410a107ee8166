// TEST INFRASTRUCTURE ONLY -- fiber scheduler behind tests/hipemu/hip/hip_runtime.h.
// Fibers are switched with a minimal x86-64 stack switch (callee-saved registers only): glibc's
// swapcontext issues two sigprocmask system calls per switch, which made MFMA-heavy kernels
// (two wave rendezvous per emulated instruction) impractically slow to model.
#include <hip/hip_runtime.h>

uint3_emu threadIdx, blockIdx;
dim3 blockDim, gridDim;

extern "C" void hipemu_switch(void** save_sp, void* load_sp);
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

Globals g;
static unsigned long long g_progress = 0;
static void* g_main_sp = nullptr;
static std::vector<void*> g_sp;

static void set_tid(int i) {
    g.cur = i;
    threadIdx = g.tid[i];
}

void yield_() {
    int me = g.cur;
    hipemu_switch(&g_sp[me], g_main_sp);
    set_tid(me);
}

void block_barrier() {
    unsigned gen = g.barrier_gen;
    g.barrier_arrived++;
    if (g.barrier_arrived >= g.alive) {
        g.barrier_arrived = 0;
        g.barrier_gen++;
        g_progress++;
        return;
    }
    while (g.barrier_gen == gen) yield_();
}

void wave_barrier() {
    Wave& w = g.waves[g.cur >> 6];
    unsigned gen = w.gen;
    w.arrived++;
    if (w.arrived >= w.nlanes) {
        w.arrived = 0;
        w.gen++;
        g_progress++;
        return;
    }
    while (w.gen == gen) yield_();
}

static void trampoline() {
    g.body();
    g.fibers[g.cur].done = true;
    g.alive--;
    g_progress++;
    // a thread that exits early must not leave a block barrier hanging
    if (g.alive > 0 && g.barrier_arrived >= g.alive) {
        g.barrier_arrived = 0;
        g.barrier_gen++;
    }
    hipemu_switch(&g_sp[g.cur], g_main_sp);
    abort();   // a finished fiber is never resumed
}

static void* make_stack(char* stack) {
    uintptr_t top = ((uintptr_t)stack + STACK_BYTES) & ~(uintptr_t)15;
    void** sp = (void**)top;
    *--sp = nullptr;                 // fake return address of trampoline
    *--sp = (void*)trampoline;       // 'ret' target of the first switch
    for (int i = 0; i < 6; ++i) *--sp = nullptr;   // rbp rbx r12 r13 r14 r15
    return (void*)sp;
}

void launch(const std::function<void()>& body, dim3 grid, dim3 block, size_t shmem) {
    int nthreads = (int)(block.x * block.y * block.z);
    if (nthreads > 1024 || nthreads <= 0) { fprintf(stderr, "hipemu: bad block size %d\n", nthreads); abort(); }
    if ((int)g.fibers.size() < nthreads) {
        size_t old = g.fibers.size();
        g.fibers.resize(nthreads);
        g_sp.resize(nthreads);
        for (size_t i = old; i < g.fibers.size(); ++i) g.fibers[i].stack = (char*)malloc(STACK_BYTES);
    }
    if (shmem + 64 > g.dyn_cap) {
        free(g.dyn_smem);
        g.dyn_cap = shmem + 64;
        g.dyn_smem = (unsigned char*)aligned_alloc(64, (g.dyn_cap + 63) / 64 * 64);
    }
    g.body = body;
    g.nthreads = nthreads;
    blockDim = block;
    gridDim = grid;
    int nwaves = (nthreads + 63) / 64;
    g.waves.assign(nwaves, Wave());
    for (int t = 0; t < nthreads; ++t) {
        g.tid[t].x = t % block.x;
        g.tid[t].y = (t / block.x) % block.y;
        g.tid[t].z = t / (block.x * block.y);
    }
    for (unsigned bz = 0; bz < grid.z; ++bz)
        for (unsigned by = 0; by < grid.y; ++by)
            for (unsigned bx = 0; bx < grid.x; ++bx) {
                blockIdx.x = bx; blockIdx.y = by; blockIdx.z = bz;
                // poison: catches reads of unwritten LDS.  HIPEMU_LDS_FILL=<byte> changes the pattern: a kernel whose results depend on
                // it reads LDS it has not written (tests/test_emu_kernels.py)
                static const int lds_fill = getenv("HIPEMU_LDS_FILL") ? atoi(getenv("HIPEMU_LDS_FILL")) : 0xCD;
                if (shmem) memset(g.dyn_smem, lds_fill, shmem);
                g.alive = nthreads;
                g.barrier_arrived = 0;
                for (int w = 0; w < nwaves; ++w) {
                    g.waves[w].arrived = 0;
                    g.waves[w].nlanes = (w == nwaves - 1) ? nthreads - 64 * w : 64;
                }
                for (int t = 0; t < nthreads; ++t) {
                    g.fibers[t].done = false;
                    g_sp[t] = make_stack(g.fibers[t].stack);
                }
                int guard = 0;
                while (g.alive > 0) {
                    unsigned long long before = g_progress;
                    int progressed = 0;
                    for (int t = 0; t < nthreads; ++t) {
                        if (g.fibers[t].done) continue;
                        set_tid(t);
                        hipemu_switch(&g_main_sp, g_sp[t]);
                        progressed++;
                    }
                    if (!progressed) break;
                    guard = (g_progress == before) ? guard + 1 : 0;
                    if (guard > 4) { fprintf(stderr, "hipemu: deadlock in block (%u,%u,%u)\n", bx, by, bz); abort(); }
                }
            }
}

}  // namespace hipemu
