// TEST INFRASTRUCTURE — the wave machine behind tests/hostsim/hip/hip_runtime.h (see there).  One workgroup at a time; its work-items
// are fibers on private stacks, scheduled wave by wave: the lanes of a wave run until each of them waits at a wave-level operation,
// at __syncthreads, or has returned; then the pending operation is resolved over the lanes that wait at it — the lanes the
// hardware would have enabled in EXEC — and those lanes run on.  When lanes of one wave wait at DIFFERENT operations (divergent
// control flow) the operation at the lowest code address goes first, which is the order structured control flow re-converges in
// (loop bodies before loop exits, a branch before the join).  Reads of a lane that is not taking part (readlane / shuffle from a
// disabled lane, which on the hardware returns whatever the register holds) return the reader's own value and are counted:
// hostsim_anomalies() — the tests assert 0.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <sys/mman.h>
#include <vector>

extern "C" void hostsim_switch(void** save_sp, void* load_sp);
asm(R"(
.text
.globl hostsim_switch
.type hostsim_switch,@function
hostsim_switch:
    pushq %rbp
    pushq %rbx
    pushq %r12
    pushq %r13
    pushq %r14
    pushq %r15
    subq $8, %rsp
    stmxcsr (%rsp)
    fnstcw 4(%rsp)
    movq %rsp, (%rdi)
    movq %rsi, %rsp
    ldmxcsr (%rsp)
    fldcw 4(%rsp)
    addq $8, %rsp
    popq %r15
    popq %r14
    popq %r13
    popq %r12
    popq %rbx
    popq %rbp
    ret
.size hostsim_switch,.-hostsim_switch
)");

namespace hostsim {

enum State { RUNNABLE, WAIT_OP, WAIT_BAR, DONE };
struct Fiber {
    Ident id;
    void* sp;
    int state;
    Post* post;
};

static const size_t STACK_BYTES = 256 << 10, MAX_THREADS = 1024, LDS_BYTES = 160 << 10;
static char* g_stacks = nullptr;
static char* g_lds = nullptr;
static Fiber g_fib[MAX_THREADS];
static void* g_sched_sp = nullptr;
static Fiber* g_run = nullptr;
static void (*g_fn)(void*) = nullptr;
static void* g_arg = nullptr;
static long g_anomalies = 0, g_split = 0, g_launches = 0;
Ident* cur = nullptr;

static void yield_to_scheduler() { hostsim_switch(&g_run->sp, g_sched_sp); }

static void fiber_main() {
    g_fn(g_arg);
    g_run->state = DONE;
    yield_to_scheduler();
    abort();                                            // a finished fiber is never resumed
}

void rendezvous(Post& p) {
    g_run->post = &p;
    g_run->state = WAIT_OP;
    yield_to_scheduler();
}
void barrier() {
    g_run->state = WAIT_BAR;
    yield_to_scheduler();
}
void* dyn_lds() { return g_lds; }

static void resume(Fiber* f) {
    g_run = f;
    cur = &f->id;
    f->state = RUNNABLE;
    hostsim_switch(&g_sched_sp, f->sp);
    g_run = nullptr;
    cur = nullptr;
}

static void prepare(Fiber* f, int t) {
    char* top = g_stacks + (size_t)(t + 1) * STACK_BYTES;          // 16-byte aligned
    void** s = reinterpret_cast<void**>(top);
    *--s = nullptr;                                                // the return address a call would have pushed
    *--s = reinterpret_cast<void*>(&fiber_main);                   // `ret` of the first switch jumps here
    for (int k = 0; k < 6; ++k) *--s = nullptr;                    // rbp rbx r12 r13 r14 r15
    unsigned csr[2];
    asm volatile("stmxcsr %0" : "=m"(csr[0]));
    unsigned short cw;
    asm volatile("fnstcw %0" : "=m"(cw));
    csr[1] = cw;
    --s;
    memcpy(s, csr, 8);
    f->sp = s;
    f->state = RUNNABLE;
    f->post = nullptr;
}

// ---- the wave-level operations, resolved over the lanes `in` (bit per lane) of wave `w` -----------------------------------------------
static void resolve(Fiber* w, uint64_t in) {
    Post* P[64];
    int first = -1;
    for (int l = 0; l < 64; ++l) {
        P[l] = (in >> l) & 1 ? w[l].post : nullptr;
        if (P[l] && first < 0) first = l;
    }
    const int op = P[first]->op;
    switch (op) {
    case OP_BALLOT: {
        uint64_t m = 0;
        for (int l = 0; l < 64; ++l) if (P[l] && P[l]->a) m |= 1ull << l;
        for (int l = 0; l < 64; ++l) if (P[l]) P[l]->res = m;
    } break;
    case OP_SHFL:
        for (int l = 0; l < 64; ++l) if (P[l]) {
            const int s = P[l]->i0 & 63;
            if (P[s]) P[l]->res = P[s]->a; else { P[l]->res = P[l]->a; ++g_anomalies; }
        }
        break;
    case OP_FIRST:
        for (int l = 0; l < 64; ++l) if (P[l]) P[l]->res = P[first]->a;
        break;
    case OP_DPP:
        for (int l = 0; l < 64; ++l) if (P[l]) {
            const int ctrl = P[l]->i0, row_mask = P[l]->i1, bank_mask = P[l]->i2, bound = P[l]->i3;
            const int row = l >> 4, r = l & 15;
            int src = -1;                                   // -1: no valid source lane
            bool applies = true;                            // row_bcast writes only the rows that receive
            if (ctrl >= 0x000 && ctrl <= 0x0FF) src = (l & ~3) | ((ctrl >> (2 * (l & 3))) & 3);                 // quad_perm
            else if (ctrl >= 0x101 && ctrl <= 0x10F) { const int n = ctrl & 15; if (r + n <= 15) src = l + n; }  // row_shl
            else if (ctrl >= 0x111 && ctrl <= 0x11F) { const int n = ctrl & 15; if (r - n >= 0) src = l - n; }   // row_shr
            else if (ctrl >= 0x121 && ctrl <= 0x12F) { const int n = ctrl & 15; src = (row << 4) | ((r - n) & 15); }   // row_ror
            else if (ctrl == 0x130) { if (l + 1 <= 63) src = l + 1; }                                           // wave_shl:1
            else if (ctrl == 0x134) src = (l + 1) & 63;                                                         // wave_rol:1
            else if (ctrl == 0x138) { if (l - 1 >= 0) src = l - 1; }                                            // wave_shr:1
            else if (ctrl == 0x13C) src = (l - 1) & 63;                                                         // wave_ror:1
            else if (ctrl == 0x140) src = (row << 4) | (15 - r);                                                // row_mirror
            else if (ctrl == 0x141) src = (l & ~7) | (7 - (l & 7));                                             // row_half_mirror
            else if (ctrl == 0x142) { if (row >= 1) src = ((row - 1) << 4) | 15; else applies = false; }         // row_bcast:15
            else if (ctrl == 0x143) { if (row >= 2) src = 31; else applies = false; }                            // row_bcast:31
            else { fprintf(stderr, "hostsim: DPP control 0x%x is not modelled\n", ctrl); abort(); }
            const bool enabled = ((row_mask >> row) & 1) && ((bank_mask >> ((l >> 2) & 3)) & 1) && applies;
            uint32_t v = (uint32_t)P[l]->b;                 // disabled rows / banks keep `old`
            if (enabled) {
                if (src >= 0 && P[src]) v = (uint32_t)P[src]->a;
                else if (bound) v = 0;                      // bound_ctrl:0 -> zero for an invalid source
            }
            P[l]->res = v;
        }
        break;
    case OP_MFMA16X16X4F32: {
        // v_mfma_f32_16x16x4_f32: A[i][k] in lane 16 k + i, B[k][j] in lane 16 k + j, D[4 (l / 16) + r][l % 16] in register r of lane l
        if (in != ~0ull) { fprintf(stderr, "hostsim: MFMA with EXEC != all ones\n"); abort(); }
        for (int l = 0; l < 64; ++l) {
            const int j = l & 15;
            for (int r = 0; r < 4; ++r) {
                const int i = 4 * (l >> 4) + r;
                float acc = P[l]->f[2 + r];
                for (int k = 0; k < 4; ++k) acc = fmaf(P[16 * k + i]->f[0], P[16 * k + j]->f[1], acc);
                P[l]->fres[r] = acc;
            }
        }
    } break;
    case OP_MFMA16X16X32BF16: {
        // v_mfma_f32_16x16x32_bf16: A[i][8 g + k] in element k of lane 16 g + i, B[8 g + k][j] in element k of lane 16 g + j,
        // D[4 (l / 16) + r][l % 16] in register r of lane l; products of bf16 values are exact in fp32, the sum is formed in fp32
        if (in != ~0ull) { fprintf(stderr, "hostsim: MFMA with EXEC != all ones\n"); abort(); }
        for (int l = 0; l < 64; ++l) {
            const int j = l & 15;
            for (int r = 0; r < 4; ++r) {
                const int i = 4 * (l >> 4) + r;
                float acc = P[l]->f[2 + r];
                for (int g = 0; g < 4; ++g)
                    for (int k = 0; k < 8; ++k) acc = fmaf(P[16 * g + i]->ext[k], P[16 * g + j]->ext[8 + k], acc);
                P[l]->fres[r] = acc;
            }
        }
    } break;
    case OP_WAVE_BARRIER: break;
    default: abort();
    }
}

// HOSTSIM_LANE_ORDER=reverse | shuffle:<seed> runs the lanes of a wave from 63 down to 0 / in a fresh pseudo-random order between
// rendezvous; HOSTSIM_WAVE_ORDER=reverse | shuffle:<seed> does the same with the waves of a workgroup between barriers.  Results
// must not depend on either: a kernel whose lanes hand data to each other through memory without a wave-level operation in
// between, or whose waves do so without __syncthreads, shows up as a difference.
struct Order {
    int mode = 0;                                       // 0 ascending, 1 reverse, 2 shuffle
    uint64_t state = 0;
    explicit Order(const char* env) {
        const char* v = getenv(env);
        if (!v) return;
        if (!strcmp(v, "reverse")) mode = 1;
        else if (!strncmp(v, "shuffle:", 8)) { mode = 2; state = 0x9E3779B97F4A7C15ull ^ strtoull(v + 8, nullptr, 10); }
    }
    void fill(int* idx, int n) {
        for (int k = 0; k < n; ++k) idx[k] = mode == 1 ? n - 1 - k : k;
        if (mode == 2)
            for (int k = n - 1; k > 0; --k) {
                state = state * 6364136223846793005ull + 1442695040888963407ull;
                const int j = (int)((state >> 33) % (uint64_t)(k + 1));
                const int t = idx[k]; idx[k] = idx[j]; idx[j] = t;
            }
    }
};
static Order g_lane_order("HOSTSIM_LANE_ORDER"), g_wave_order("HOSTSIM_WAVE_ORDER");

static void run_wave(Fiber* w, int n_lanes) {
    for (;;) {
        int idx[64];
        g_lane_order.fill(idx, n_lanes);
        for (int k = 0; k < n_lanes; ++k)
            if (w[idx[k]].state == RUNNABLE) resume(&w[idx[k]]);
        // every lane now waits or is done
        const void* best = nullptr;
        int best_tag = 0, best_op = 0, groups = 0;
        for (int l = 0; l < n_lanes; ++l)
            if (w[l].state == WAIT_OP) {
                const Post* p = w[l].post;
                if (!best || p->site < best || (p->site == best && p->tag < best_tag)) { best = p->site; best_tag = p->tag; best_op = p->op; }
            }
        if (!best) return;
        uint64_t in = 0;
        for (int l = 0; l < n_lanes; ++l)
            if (w[l].state == WAIT_OP) {
                if (w[l].post->site == best && w[l].post->tag == best_tag && w[l].post->op == best_op) in |= 1ull << l;
                else ++groups;
            }
        if (groups) ++g_split;
        // lanes beyond n_lanes do not exist: resolve() sees them as disabled
        resolve(w, in);
        for (int l = 0; l < n_lanes; ++l)
            if ((in >> l) & 1) w[l].state = RUNNABLE;
    }
}

void launch_impl(dim3 grid, dim3 block, size_t shmem, void (*fn)(void*), void* arg) {
    const size_t nt = (size_t)block.x * block.y * block.z;
    if (nt == 0 || nt > MAX_THREADS || shmem > LDS_BYTES) { fprintf(stderr, "hostsim: bad launch (%zu threads, %zu B LDS)\n", nt, shmem); abort(); }
    if (!g_stacks) {
        g_stacks = (char*)mmap(nullptr, STACK_BYTES * MAX_THREADS, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
        g_lds = (char*)aligned_alloc(256, LDS_BYTES);
        if (g_stacks == MAP_FAILED || !g_lds) { fprintf(stderr, "hostsim: no memory for the fiber stacks\n"); abort(); }
        memset(g_lds, 0x5A, LDS_BYTES);
    }
    ++g_launches;
    g_fn = fn;
    g_arg = arg;
    const int n_waves = (int)((nt + 63) / 64);
    // HOSTSIM_BLOCK_ORDER=reverse: the workgroups of a launch from the last to the first (no kernel may rely on the dispatch order)
    static const bool block_reverse = getenv("HOSTSIM_BLOCK_ORDER") && !strcmp(getenv("HOSTSIM_BLOCK_ORDER"), "reverse");
    for (unsigned bz0 = 0; bz0 < grid.z; ++bz0)
        for (unsigned by0 = 0; by0 < grid.y; ++by0)
            for (unsigned bx0 = 0; bx0 < grid.x; ++bx0) {
                const unsigned bx = block_reverse ? grid.x - 1 - bx0 : bx0, by = block_reverse ? grid.y - 1 - by0 : by0,
                               bz = block_reverse ? grid.z - 1 - bz0 : bz0;
                for (size_t t = 0; t < nt; ++t) {
                    Fiber* f = &g_fib[t];
                    f->id.tid = uint3{(unsigned)(t % block.x), (unsigned)(t / block.x % block.y), (unsigned)(t / ((size_t)block.x * block.y))};
                    f->id.bid = uint3{bx, by, bz};
                    f->id.bdim = block;
                    f->id.gdim = grid;
                    f->id.lane = (int)(t & 63);
                    prepare(f, (int)t);
                }
                for (;;) {
                    int widx[MAX_THREADS / 64];
                    g_wave_order.fill(widx, n_waves);
                    for (int k = 0; k < n_waves; ++k) {
                        const int w = widx[k];
                        run_wave(&g_fib[w * 64], (int)std::min<size_t>(64, nt - (size_t)w * 64));
                    }
                    size_t n_bar = 0, n_done = 0;
                    for (size_t t = 0; t < nt; ++t) { n_bar += g_fib[t].state == WAIT_BAR; n_done += g_fib[t].state == DONE; }
                    if (n_done == nt) break;
                    if (n_bar + n_done != nt) { fprintf(stderr, "hostsim: scheduler inconsistency\n"); abort(); }
                    for (size_t t = 0; t < nt; ++t) if (g_fib[t].state == WAIT_BAR) g_fib[t].state = RUNNABLE;
                }
            }
}

}  // namespace hostsim

extern "C" long hostsim_anomalies() { return hostsim::g_anomalies; }
extern "C" long hostsim_split_waves() { return hostsim::g_split; }
extern "C" long hostsim_launches() { return hostsim::g_launches; }
extern "C" void hostsim_reset_counters() { hostsim::g_anomalies = hostsim::g_split = hostsim::g_launches = 0; }
