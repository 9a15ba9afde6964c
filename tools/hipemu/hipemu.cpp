// hipemu runtime: fibers + rendezvous primitives.  TEST-ONLY (see hip/hip_runtime.h).
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <vector>

namespace hipemu {

// ---- minimal x86-64 context switch (callee-saved registers + stack pointer) -------------------------------
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

static constexpr size_t kStack = 256 * 1024;

struct Fiber {
    void* sp = nullptr;
    char* stack = nullptr;
    bool done = false;
    ThreadCtx ctx;
};

struct WaveState {
    int arrived = 0, gen = 0, released = 0, rgen = 0, lanes = 0;
    std::vector<char> buf;
};

struct Worker {
    std::vector<Fiber> fibers;
    std::vector<WaveState> waves;
    void* sched_sp = nullptr;
    int cur = -1, nthreads = 0, alive = 0;
    int bar_arrived = 0, bar_gen = 0;
    const std::function<void()>* body = nullptr;
    std::vector<char> smem;
};

static thread_local Worker* tl_worker = nullptr;
static thread_local ThreadCtx* tl_ctx = nullptr;

ThreadCtx& tctx() { return *tl_ctx; }

static void yield_to_sched() {
    Worker* w = tl_worker;
    Fiber& f = w->fibers[w->cur];
    hipemu_switch(&f.sp, w->sched_sp);
}

static void fiber_entry() {
    Worker* w = tl_worker;
    (*w->body)();
    Fiber& f = w->fibers[w->cur];
    f.done = true;
    w->alive--;
    yield_to_sched();
    abort();
}

void block_barrier() {
    Worker* w = tl_worker;
    int gen = w->bar_gen;
    if (++w->bar_arrived == w->nthreads) {
        w->bar_arrived = 0;
        w->bar_gen++;
        return;
    }
    while (w->bar_gen == gen) {
        if (w->alive < w->nthreads && w->bar_arrived >= w->alive) {
            fprintf(stderr, "hipemu: __syncthreads() reached by %d threads but %d exited early (divergent barrier)\n",
                    w->bar_arrived, w->nthreads - w->alive);
            abort();
        }
        yield_to_sched();
    }
}

const char* wave_gather(const void* src, size_t bytes) {
    Worker* w = tl_worker;
    ThreadCtx& c = *tl_ctx;
    WaveState& ws = w->waves[c.wave];
    if (ws.buf.size() < 64 * bytes) ws.buf.resize(64 * bytes);
    std::memcpy(ws.buf.data() + (size_t)c.lane * bytes, src, bytes);
    int gen = ws.gen;
    if (++ws.arrived == ws.lanes) {
        ws.arrived = 0;
        ws.gen++;
    } else {
        while (ws.gen == gen) yield_to_sched();
    }
    return ws.buf.data();
}

void wave_release() {
    Worker* w = tl_worker;
    WaveState& ws = w->waves[tl_ctx->wave];
    int gen = ws.rgen;
    if (++ws.released == ws.lanes) {
        ws.released = 0;
        ws.rgen++;
    } else {
        while (ws.rgen == gen) yield_to_sched();
    }
}

static void run_block(Worker& w, dim3 grid, dim3 block, dim3 bid, size_t shmem, const std::function<void()>& body) {
    int n = (int)(block.x * block.y * block.z);
    if ((int)w.fibers.size() < n) {
        size_t old = w.fibers.size();
        w.fibers.resize(n);
        for (size_t i = old; i < (size_t)n; ++i) w.fibers[i].stack = (char*)aligned_alloc(64, kStack);
    }
    w.nthreads = n;
    w.alive = n;
    w.bar_arrived = 0;
    w.body = &body;
    int nw = (n + 63) / 64;
    if ((int)w.waves.size() < nw) w.waves.resize(nw);
    for (int i = 0; i < nw; ++i) {
        w.waves[i].arrived = w.waves[i].released = 0;
        w.waves[i].lanes = (i == nw - 1) ? n - 64 * i : 64;
    }
    if (w.smem.size() < shmem + 64) w.smem.resize(shmem + 64);
    char* smem = (char*)(((uintptr_t)w.smem.data() + 63) & ~(uintptr_t)63);
    for (int i = 0; i < n; ++i) {
        Fiber& f = w.fibers[i];
        f.done = false;
        f.ctx.tid = dim3(i % block.x, (i / block.x) % block.y, i / (block.x * block.y));
        f.ctx.bid = bid;
        f.ctx.bdim = block;
        f.ctx.gdim = grid;
        f.ctx.dyn_smem = smem;
        f.ctx.linear = i;
        f.ctx.lane = i & 63;
        f.ctx.wave = i >> 6;
        // initial frame: 6 callee-saved slots + return address = fiber_entry; entry must see rsp % 16 == 8
        uintptr_t top = ((uintptr_t)f.stack + kStack) & ~(uintptr_t)15;
        void** sp = (void**)(top - 8);   // slot that `ret` leaves rsp pointing past
        *--sp = (void*)&fiber_entry;     // return address
        for (int r = 0; r < 6; ++r) *--sp = nullptr;
        f.sp = (void*)sp;
    }
    while (w.alive > 0) {
        for (int i = 0; i < n; ++i) {
            Fiber& f = w.fibers[i];
            if (f.done) continue;
            w.cur = i;
            tl_ctx = &f.ctx;
            hipemu_switch(&w.sched_sp, f.sp);
        }
    }
    tl_ctx = nullptr;
}

static int n_workers() {
    static int n = [] {
        const char* e = getenv("HIPEMU_THREADS");
        int v = e ? atoi(e) : (int)std::thread::hardware_concurrency();
        return v < 1 ? 1 : v;
    }();
    return n;
}

// persistent worker pool: fiber stacks are allocated once per worker and reused across launches
struct Job {
    dim3 grid, block;
    size_t shmem = 0, total = 0;
    const std::function<void()>* body = nullptr;
    std::atomic<size_t> next{0};
};
struct Pool {
    std::vector<std::thread> threads;
    std::mutex m;
    std::condition_variable cv, cv_done;
    Job* job = nullptr;
    uint64_t epoch = 0;
    int running = 0;
    Pool() {
        int n = n_workers();
        for (int i = 0; i < n; ++i) threads.emplace_back([this] { loop(); });
    }
    void loop() {
        Worker worker;
        tl_worker = &worker;
        uint64_t seen = 0;
        for (;;) {
            Job* j;
            {
                std::unique_lock<std::mutex> lk(m);
                cv.wait(lk, [&] { return epoch != seen; });
                seen = epoch;
                j = job;
            }
            for (;;) {
                size_t b = j->next.fetch_add(1);
                if (b >= j->total) break;
                dim3 bid((unsigned)(b % j->grid.x), (unsigned)((b / j->grid.x) % j->grid.y),
                         (unsigned)(b / ((size_t)j->grid.x * j->grid.y)));
                run_block(worker, j->grid, j->block, bid, j->shmem, *j->body);
            }
            {
                std::lock_guard<std::mutex> lk(m);
                if (--running == 0) cv_done.notify_all();
            }
        }
    }
    void run(Job& j) {
        std::unique_lock<std::mutex> lk(m);
        job = &j;
        running = (int)threads.size();
        ++epoch;
        cv.notify_all();
        cv_done.wait(lk, [&] { return running == 0; });
    }
};

void launch(dim3 grid, dim3 block, size_t shmem, const std::function<void()>& body) {
    static Pool* pool = new Pool();  // leaked on purpose: workers live for the process
    static std::mutex launch_mutex;
    std::lock_guard<std::mutex> g(launch_mutex);
    Job j;
    j.grid = grid;
    j.block = block;
    j.shmem = shmem;
    j.total = (size_t)grid.x * grid.y * grid.z;
    j.body = &body;
    if (j.total == 0) return;
    pool->run(j);
}

}  // namespace hipemu

// ---- runtime API ------------------------------------------------------------------------------------------
struct hipemu_event { std::chrono::steady_clock::time_point t; };
const char* hipGetErrorString(hipError_t e) { return e == hipSuccess ? "hipSuccess" : "hipemu error"; }
hipError_t hipGetLastError() { return hipSuccess; }
hipError_t hipMalloc(void** p, size_t n) {
    *p = aligned_alloc(256, (n + 255) & ~(size_t)255);
    return *p ? hipSuccess : hipErrorOutOfMemory;
}
hipError_t hipFree(void* p) { free(p); return hipSuccess; }
hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { std::memcpy(d, s, n); return hipSuccess; }
hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { std::memcpy(d, s, n); return hipSuccess; }
hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { std::memset(d, v, n); return hipSuccess; }
hipError_t hipMemset(void* d, int v, size_t n) { std::memset(d, v, n); return hipSuccess; }
hipError_t hipDeviceSynchronize() { return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
hipError_t hipSetDevice(int) { return hipSuccess; }
hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
hipError_t hipEventCreate(hipEvent_t* e) { *e = new hipemu_event(); return hipSuccess; }
hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t e, hipStream_t) { e->t = std::chrono::steady_clock::now(); return hipSuccess; }
hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) {
    *ms = std::chrono::duration<float, std::milli>(b->t - a->t).count();
    return hipSuccess;
}
