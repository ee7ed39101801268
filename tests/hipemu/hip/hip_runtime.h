// TEST-ONLY host emulation of the small slice of the HIP kernel language that
// diffmvs_amd/csrc uses.  It lets `python -m pytest -m "not gpu"` execute the *same* kernel
// sources on CPU threads (tests/hipemu/build.py compiles csrc/*.hip with g++ -I tests/hipemu),
// so index arithmetic, tiling and epilogues are checked against the oracle in a container
// that has no GPU.  It is NOT a product backend: nothing under diffmvs_amd/ or models/
// references it, the product loader only ever opens libdmvs_hip.so (gfx950 code object).
//
// Model: one fiber (own stack + hand-rolled context switch) per GPU thread, the fibers of a block run round-robin on one
// OS thread; __syncthreads() and the wave shuffles are "yield until the scheduler comes
// round again", which is a block-wide barrier as long as every live thread of the block
// executes the same sequence of them (the same rule the hardware imposes on barriers).
#pragma once
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <thread>
#include <vector>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __shared__ static thread_local

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct float2 { float x, y; };
struct float3 { float x, y, z; };
struct alignas(16) float4 { float x, y, z, w; };
struct alignas(8) uint2 { unsigned x, y; };
struct alignas(16) uint4 { unsigned x, y, z, w; };
static inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
static inline float3 make_float3(float x, float y, float z) { return float3{x, y, z}; }
static inline float2 make_float2(float x, float y) { return float2{x, y}; }

typedef void* hipStream_t;
typedef int hipError_t;
enum { hipSuccess = 0 };
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { memset(p, v, n); return hipSuccess; }

// Context switch: six callee-saved registers + the stack pointer (x86-64 SysV).  swapcontext() would cost two
// sigprocmask system calls per switch, and a block of 256 fibers switches ~10^5 times per kernel.
extern "C" void hipemu_switch(void** save_sp, void* load_sp);
asm(R"(
.text
.weak hipemu_switch
.hidden hipemu_switch
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
.size hipemu_switch, .-hipemu_switch
)");

#include <sys/mman.h>

#include <mutex>

namespace hipemu {
struct Fiber {
    void* sp;
    bool done;
};
constexpr size_t kStack = 256 * 1024;
constexpr int kMaxThreads = 1024;

// fiber stacks: one lazily-committed slab per concurrently running worker, recycled across launches
struct StackPool {
    std::mutex mu;
    std::vector<char*> free_;
    char* get() {
        {
            std::lock_guard<std::mutex> l(mu);
            if (!free_.empty()) {
                char* p = free_.back();
                free_.pop_back();
                return p;
            }
        }
        void* p = mmap(nullptr, kStack * kMaxThreads, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
        if (p == MAP_FAILED) {
            perror("hipemu: mmap");
            abort();
        }
        return (char*)p;
    }
    void put(char* p) {
        std::lock_guard<std::mutex> l(mu);
        free_.push_back(p);
    }
};
inline StackPool& stack_pool() {
    static StackPool* p = new StackPool;
    return *p;
}

struct Worker {
    void* main_sp = nullptr;
    std::vector<Fiber> fibers;
    char* stacks = nullptr;
    std::vector<uint64_t> slots;   // shuffle exchange, one per thread of the block
    // rendezvous of the cross-lane exchanges, one counting barrier per group and group kind (0 = waves: the shuffles,
    // 1 = quads: DPP quad_perm)
    struct GroupBar { int arrived = 0, live = 0; unsigned gen = 0; };
    std::vector<GroupBar> gbar[2];
    std::vector<float> slots_a, slots_b;   // MFMA operand exchange
    int current = -1;
    int alive = 0;                 // fibers of the running block that have not returned yet
    int bar_count = 0;             // arrivals at the current __syncthreads()
    unsigned bar_gen = 0;          // barrier generation
    const std::function<void()>* body = nullptr;
};
inline thread_local Worker* g_worker = nullptr;
inline thread_local dim3 g_threadIdx, g_blockIdx, g_blockDim, g_gridDim;

inline void set_tid(int t) {
    g_threadIdx.x = t % g_blockDim.x;
    g_threadIdx.y = (t / g_blockDim.x) % g_blockDim.y;
    g_threadIdx.z = t / (g_blockDim.x * g_blockDim.y);
}
inline void yield() {
    Worker* w = g_worker;
    hipemu_switch(&w->fibers[w->current].sp, w->main_sp);
}
__attribute__((noinline)) inline void trampoline() {
    Worker* w = g_worker;
    (*w->body)();
    w = g_worker;
    w->fibers[w->current].done = true;
    yield();
    abort();   // a finished fiber is never resumed
}
inline int linear_tid() { return g_threadIdx.x + g_blockDim.x * (g_threadIdx.y + g_blockDim.y * g_threadIdx.z); }

inline void run_block(Worker& w, int nthreads) {
    for (int k = 0; k < 2; ++k) {
        const int gs = k == 0 ? 64 : 4;
        for (size_t g = 0; g < w.gbar[k].size(); ++g) {
            w.gbar[k][g].arrived = 0;
            w.gbar[k][g].live = std::min(gs, nthreads - (int)g * gs);
        }
    }
    for (int t = 0; t < nthreads; ++t) {
        Fiber& f = w.fibers[t];
        f.done = false;
        // initial frame: six zeroed callee-saved registers, then `ret` into trampoline with rsp % 16 == 8 as at a call
        uintptr_t top = (uintptr_t)(w.stacks + size_t(t + 1) * kStack) & ~uintptr_t(15);
        void** slot = (void**)(top - 16);
        slot[0] = (void*)&trampoline;
        slot[1] = nullptr;
        void** sp = slot - 6;
        for (int i = 0; i < 6; ++i) sp[i] = nullptr;
        f.sp = sp;
    }
    w.alive = nthreads;
    w.bar_count = 0;
    while (w.alive > 0) {
        for (int t = 0; t < nthreads; ++t) {
            if (w.fibers[t].done) continue;
            w.current = t;
            set_tid(t);
            hipemu_switch(&w.main_sp, w.fibers[t].sp);
            if (w.fibers[t].done) {
                --w.alive;
                --w.gbar[0][t >> 6].live;
                --w.gbar[1][t >> 2].live;
            }
        }
    }
}

inline void launch(dim3 grid, dim3 block, size_t, const std::function<void()>& body) {
    const long nblocks = long(grid.x) * grid.y * grid.z;
    const int nthreads = int(block.x * block.y * block.z);
    if (nblocks == 0 || nthreads == 0) return;
    if (nthreads > kMaxThreads) abort();
    std::atomic<long> next{0};
    auto work = [&]() {
        Worker w;
        w.fibers.resize(nthreads);
        w.stacks = stack_pool().get();
        w.slots.resize(nthreads);
        w.gbar[0].resize((nthreads + 63) / 64);
        w.gbar[1].resize((nthreads + 3) / 4);
        w.slots_a.resize(nthreads);
        w.slots_b.resize(nthreads);
        w.body = &body;
        g_worker = &w;
        g_blockDim = block;
        g_gridDim = grid;
        for (;;) {
            long b = next.fetch_add(1);
            if (b >= nblocks) break;
            g_blockIdx.x = unsigned(b % grid.x);
            g_blockIdx.y = unsigned((b / grid.x) % grid.y);
            g_blockIdx.z = unsigned(b / (long(grid.x) * grid.y));
            run_block(w, nthreads);
        }
        g_worker = nullptr;
        stack_pool().put(w.stacks);
    };
    unsigned nw = std::min<long>(std::max(1u, std::thread::hardware_concurrency()), nblocks);
    std::vector<std::thread> pool;
    for (unsigned i = 1; i < nw; ++i) pool.emplace_back(work);
    work();
    for (auto& t : pool) t.join();
}

// Cross-lane exchange inside a group of `gsize` adjacent threads (64: the wave-wide shuffles; 4: DPP quad_perm).  The members
// of a group rendezvous explicitly -- every live member has posted its value before anyone reads, everyone has read before
// anyone posts the next one -- because the groups of a block do not run in lockstep here: the quad kernels loop a
// data-dependent number of times per quad, and unlike a hardware wave the fibers do not reconverge afterwards.
inline void group_barrier(Worker::GroupBar& gb) {
    const unsigned gen = gb.gen;
    ++gb.arrived;
    for (;;) {
        if (gb.gen != gen) return;
        if (gb.arrived >= gb.live) {      // (re-checked while waiting: members may return meanwhile)
            gb.arrived = 0;
            ++gb.gen;
            return;
        }
        yield();
    }
}
template <typename T>
inline T shfl_from(T v, int src_lane_abs, int gsize = 64) {
    Worker* w = g_worker;
    const int t = linear_tid();
    Worker::GroupBar& gb = gsize == 64 ? w->gbar[0][t >> 6] : w->gbar[1][t >> 2];
    uint64_t raw = 0;
    memcpy(&raw, &v, sizeof(T));
    w->slots[t] = raw;
    group_barrier(gb);                    // every live member has posted
    uint64_t got = w->slots[src_lane_abs];
    group_barrier(gb);                    // every live member has read: the slots may be overwritten
    T r;
    memcpy(&r, &got, sizeof(T));
    return r;
}
}  // namespace hipemu

// Real LDS holds whatever the previous workgroup left there; a `static thread_local` array starts as zeros, which hides reads of
// positions a kernel never wrote (e.g. padding it forgot to clear).  Kernels call DMVS_LDS_POISON(array) (csrc/dmvs_lds_poison.h)
// right after declaring their LDS: the block's first fiber -- fibers start in order and run until their first barrier / shuffle --
// fills it with 0xff bytes (NaN as float / double, -1 as int) before any other fiber of the block has started.
inline void hipemu_poison_lds(void* p, size_t bytes) {
    if (hipemu::g_threadIdx.x == 0 && hipemu::g_threadIdx.y == 0 && hipemu::g_threadIdx.z == 0) memset(p, 0xff, bytes);
}
#define threadIdx hipemu::g_threadIdx
#define blockIdx hipemu::g_blockIdx
#define blockDim hipemu::g_blockDim
#define gridDim hipemu::g_gridDim
#define hipLaunchKernelGGL(k, g, b, sh, st, ...) \
    hipemu::launch((g), (b), (sh), std::function<void()>([=]() { k(__VA_ARGS__); }))

// A real barrier (arrival count + generation), like s_barrier: fibers may reach it after different numbers of yields -- the
// quad kernels run DPP exchanges (two yields each) inside per-quad loops of data-dependent length.  Fibers that have returned
// no longer count, as on the hardware.
static inline void __syncthreads() {
    hipemu::Worker* w = hipemu::g_worker;
    const unsigned gen = w->bar_gen;
    ++w->bar_count;
    for (;;) {
        if (w->bar_gen != gen) break;
        if (w->bar_count >= w->alive) {
            w->bar_count = 0;
            ++w->bar_gen;
            break;
        }
        hipemu::yield();
    }
}
template <typename T>
static inline T __shfl_down(T v, unsigned delta, int width = 64) {
    int t = hipemu::linear_tid();
    int lane = t & 63, base = t & ~63;
    int src = lane + int(delta);
    if ((lane & (width - 1)) + int(delta) >= width) src = lane;
    return hipemu::shfl_from(v, base + src);
}
template <typename T>
static inline T __shfl_xor(T v, int mask, int width = 64) {
    int t = hipemu::linear_tid();
    int lane = t & 63, base = t & ~63;
    (void)width;
    return hipemu::shfl_from(v, base + (lane ^ mask));
}
template <typename T>
static inline T __shfl(T v, int src, int width = 64) {
    int t = hipemu::linear_tid();
    int lane = t & 63, base = t & ~63;
    int grp = lane & ~(width - 1);
    return hipemu::shfl_from(v, base + grp + (src & (width - 1)));
}

template <typename T>
static inline T hipemu_atomic_add(T* addr, T val) {
    using U = typename std::conditional<sizeof(T) == 4, uint32_t, uint64_t>::type;
    U* p = reinterpret_cast<U*>(addr);
    U old = __atomic_load_n(p, __ATOMIC_RELAXED);
    for (;;) {
        T cur;
        memcpy(&cur, &old, sizeof(T));
        T nv = cur + val;
        U nraw;
        memcpy(&nraw, &nv, sizeof(T));
        if (__atomic_compare_exchange_n(p, &old, nraw, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) return cur;
    }
}
static inline float atomicAdd(float* a, float v) { return hipemu_atomic_add(a, v); }
static inline double atomicAdd(double* a, double v) { return hipemu_atomic_add(a, v); }
static inline int atomicAdd(int* a, int v) { return __atomic_fetch_add(a, v, __ATOMIC_RELAXED); }
static inline unsigned long long atomicAdd(unsigned long long* a, unsigned long long v) { return __atomic_fetch_add(a, v, __ATOMIC_RELAXED); }

// clang's ext_vector_type spelled for g++ (kernels only use 4 x float vectors)
#define ext_vector_type(N) vector_size((N) * 4)
typedef float hipemu_f32x4 __attribute__((vector_size(16)));
// v_mfma_f32_16x16x4_f32 (cdna_hip_programming.md section 3): A[i = lane&15][k = lane>>4], B[k = lane>>4][j = lane&15],
// D[row = 4*(lane>>4) + r][col = lane&15], k-ordered fma chain.
static inline hipemu_f32x4 __builtin_amdgcn_mfma_f32_16x16x4f32(float a, float b, hipemu_f32x4 c, int, int, int) {
    hipemu::Worker* w = hipemu::g_worker;
    const int t = hipemu::linear_tid(), lane = t & 63, base = t & ~63;
    w->slots_a[t] = a;
    w->slots_b[t] = b;
    hipemu::yield();
    const int col = lane & 15;
    for (int r = 0; r < 4; ++r) {
        const int row = 4 * (lane >> 4) + r;
        float acc = c[r];
        for (int k = 0; k < 4; ++k) acc = fmaf(w->slots_a[base + k * 16 + row], w->slots_b[base + k * 16 + col], acc);
        c[r] = acc;
    }
    hipemu::yield();
    return c;
}
// v_mfma_f32_16x16x32_bf16: A[i = lane&15][k = 8*(lane>>4) + j], B[k = 8*(lane>>4) + j][n = lane&15], j = 0..7 (raw bf16 bit
// patterns); D[row = 4*(lane>>4) + r][col = lane&15]; products exact in fp32, fp32 accumulation (the summation order inside
// the instruction is not architecturally specified: k-ascending here)
typedef short hipemu_s16x8 __attribute__((vector_size(16)));
static inline hipemu_f32x4 hipemu_mfma_f32_16x16x32_bf16(hipemu_s16x8 a, hipemu_s16x8 b, hipemu_f32x4 c) {
    hipemu::Worker* w = hipemu::g_worker;
    const int t = hipemu::linear_tid(), lane = t & 63, base = t & ~63;
    static thread_local std::vector<float> sa, sb;
    if (sa.size() < w->fibers.size() * 8) { sa.resize(w->fibers.size() * 8); sb.resize(w->fibers.size() * 8); }
    for (int j = 0; j < 8; ++j) {
        uint32_t ua = (uint32_t)(uint16_t)a[j] << 16, ub = (uint32_t)(uint16_t)b[j] << 16;
        memcpy(&sa[t * 8 + j], &ua, 4);
        memcpy(&sb[t * 8 + j], &ub, 4);
    }
    hipemu::yield();
    const int col = lane & 15;
    for (int r = 0; r < 4; ++r) {
        const int row = 4 * (lane >> 4) + r;
        float acc = c[r];
        for (int k = 0; k < 32; ++k) acc += sa[(base + (k >> 3) * 16 + row) * 8 + (k & 7)] * sb[(base + (k >> 3) * 16 + col) * 8 + (k & 7)];
        c[r] = acc;
    }
    hipemu::yield();
    return c;
}
// LDS-DMA (global_load_lds_dword / _dwordx4): per-lane global source, LDS destination =
// wave-uniform base + lane * size.  Synchronous here; the kernels' barriers make that equivalent.
// The kernels only ever issue the 16-byte form with 16-byte aligned global sources and LDS destinations (their dispatch rules check
// the tensors for it); the emulation enforces that invariant instead of relying on what a particular GPU tolerates.
static inline void __builtin_amdgcn_global_load_lds(const void* g, void* lds_base, unsigned size, int offset, int) {
    const int lane = hipemu::linear_tid() & 63;
    char* dst = static_cast<char*>(lds_base) + offset + size_t(lane) * size;
    if (size == 16 && ((reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(dst)) & 15) != 0) {
        fprintf(stderr, "hipemu: 16-byte LDS-DMA with a misaligned source %p or destination %p\n", g, (void*)dst);
        abort();
    }
    memcpy(dst, g, size);
}
// HIP's global integer min / max
static inline int min(int a, int b) { return a < b ? a : b; }
static inline int max(int a, int b) { return a > b ? a : b; }
static inline float rsqrtf(float x) { return 1.0f / sqrtf(x); }
static inline float __frcp_rn(float x) { return 1.0f / x; }
static inline float __fdividef(float a, float b) { return a / b; }
static inline void __builtin_amdgcn_sched_barrier(int) {}
static inline void __threadfence() { __atomic_thread_fence(__ATOMIC_SEQ_CST); }
static inline float __builtin_amdgcn_rcpf(float x) { return 1.0f / x; }
static inline int __mul24(int a, int b) { return a * b; }
// only ever applied to wave-uniform values (the wave index): identity under one-fiber-per-thread emulation
static inline int __builtin_amdgcn_readfirstlane(int x) { return x; }
// DPP quad_perm: lane l of every quad reads lane (ctrl >> 2*(l&3)) & 3 of the same quad
static inline int hipemu_quad_perm(int v, int ctrl) {
    const int t = hipemu::linear_tid(), lane = t & 63;
    return hipemu::shfl_from(v, (t & ~63) + ((lane & ~3) | ((ctrl >> (2 * (lane & 3))) & 3)), 4);
}
#define DMVS_QUAD_PERM(v, ctrl) hipemu_quad_perm((v), (ctrl))
#define DMVS_HOST_EMULATION 1
static inline int __ffsll(long long v) { return __builtin_ffsll(v); }
static inline int __ffs(int v) { return __builtin_ffs(v); }
static inline uint32_t __float_as_uint(float f) { uint32_t v; memcpy(&v, &f, 4); return v; }
static inline float __uint_as_float(uint32_t v) { float f; memcpy(&f, &v, 4); return f; }
// IEEE binary16 <-> binary32, round to nearest even (the hardware's v_cvt_f16_f32 / v_cvt_f32_f16)
static inline uint16_t hipemu_f32_to_f16(float f) {
    uint32_t x = __float_as_uint(f);
    const uint32_t sign = (x >> 16) & 0x8000u;
    x &= 0x7fffffffu;
    if (x >= 0x7f800000u) return (uint16_t)(sign | (x > 0x7f800000u ? 0x7e00u : 0x7c00u));
    if (x >= 0x477ff000u) return (uint16_t)(sign | 0x7c00u);                     // rounds to >= 65520 -> inf
    if (x < 0x38800000u) {                                                       // subnormal half (or zero)
        if (x < 0x33000000u) return (uint16_t)sign;
        const int e = (int)(x >> 23);
        uint32_t m = (x & 0x7fffffu) | 0x800000u;
        const int shift = 126 - e;                                               // 14 .. 24
        const uint32_t half = m >> shift, rem = m & ((1u << shift) - 1u), mid = 1u << (shift - 1);
        return (uint16_t)(sign | (half + ((rem > mid || (rem == mid && (half & 1u))) ? 1u : 0u)));
    }
    uint32_t h = ((x - 0x38000000u) >> 13);
    const uint32_t rem = x & 0x1fffu;
    h += (rem > 0x1000u || (rem == 0x1000u && (h & 1u))) ? 1u : 0u;
    return (uint16_t)(sign | h);
}
static inline float hipemu_f16_to_f32(uint16_t h) {
    const uint32_t sign = (uint32_t)(h & 0x8000u) << 16, e = (h >> 10) & 0x1fu, m = h & 0x3ffu;
    if (e == 0) {
        if (m == 0) return __uint_as_float(sign);
        float v = (float)m * (1.0f / 16777216.0f);                                // m * 2^-24
        return (h & 0x8000u) ? -v : v;
    }
    if (e == 31) return __uint_as_float(sign | 0x7f800000u | (m << 13));
    return __uint_as_float(sign | ((e + 112u) << 23) | (m << 13));
}
static inline float __int_as_float(int v) { float f; memcpy(&f, &v, 4); return f; }
static inline int __float_as_int(float f) { int v; memcpy(&v, &f, 4); return v; }
#define DMVS_ORDER_AFTER(var, dep) ((void)0)
#define DMVS_LDS_BARRIER() __syncthreads()
#define DMVS_DMA_BARRIER() __syncthreads()
