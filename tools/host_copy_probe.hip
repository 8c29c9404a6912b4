// tools/host_copy_probe.hip -- what a 16.8 MB host array costs to move on this box (round 6, the numpy drop-in path):
// pageable hipMemcpy H2D / D2H, pinned async copies, hipHostRegister, and the CPU memcpy into a pinned bounce buffer
// with 1 .. 16 threads.    hipcc -O2 --offload-arch=gfx950 tools/host_copy_probe.hip -o /tmp/hcp -lpthread && /tmp/hcp
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static void par_copy(char* dst, const char* src, size_t n, int nt) {
    if (nt <= 1) { memcpy(dst, src, n); return; }
    std::vector<std::thread> th;
    const size_t per = (n / nt + 4095) & ~(size_t)4095;
    for (int i = 0; i < nt; ++i) {
        const size_t a = (size_t)i * per; if (a >= n) break;
        const size_t len = a + per > n ? n - a : per;
        th.emplace_back([=] { memcpy(dst + a, src + a, len); });
    }
    for (auto& t : th) t.join();
}
int main() {
    const size_t N = (size_t)4096 * 512 * 8;     // one L x T matrix of doubles: 16.8 MB
    char* pageable = (char*)malloc(N); memset(pageable, 1, N);
    char* pageable2 = (char*)malloc(N); memset(pageable2, 2, N);
    char *pinned, *dev;
    CHECK(hipHostMalloc((void**)&pinned, N, hipHostMallocDefault));
    memset(pinned, 3, N);
    CHECK(hipMalloc((void**)&dev, N));
    hipStream_t s; CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    auto rep = [&](const char* name, int reps, auto fn) {
        fn(); double best = 1e9, sum = 0;
        for (int i = 0; i < reps; ++i) { double t0 = now(); fn(); double t = now() - t0; best = t < best ? t : best; sum += t; }
        printf("%-52s best %.3f ms  mean %.3f ms  (%.1f GB/s best)\n", name, best * 1e3, sum / reps * 1e3, N / best / 1e9);
    };
    rep("hipMemcpy H2D pageable", 10, [&] { CHECK(hipMemcpy(dev, pageable, N, hipMemcpyHostToDevice)); });
    rep("hipMemcpy D2H pageable", 10, [&] { CHECK(hipMemcpy(pageable2, dev, N, hipMemcpyDeviceToHost)); });
    rep("hipMemcpyAsync H2D pinned + sync", 10, [&] { CHECK(hipMemcpyAsync(dev, pinned, N, hipMemcpyHostToDevice, s)); CHECK(hipStreamSynchronize(s)); });
    rep("hipMemcpyAsync D2H pinned + sync", 10, [&] { CHECK(hipMemcpyAsync(pinned, dev, N, hipMemcpyDeviceToHost, s)); CHECK(hipStreamSynchronize(s)); });
    rep("hipMemcpyAsync H2D pageable + sync", 10, [&] { CHECK(hipMemcpyAsync(dev, pageable, N, hipMemcpyHostToDevice, s)); CHECK(hipStreamSynchronize(s)); });
    rep("hipHostRegister + H2D + unregister", 5, [&] { CHECK(hipHostRegister(pageable, N, hipHostRegisterDefault)); CHECK(hipMemcpyAsync(dev, pageable, N, hipMemcpyHostToDevice, s)); CHECK(hipStreamSynchronize(s)); CHECK(hipHostUnregister(pageable)); });
    for (int nt : {1, 2, 4, 8, 16}) {
        char nm[64]; snprintf(nm, sizeof nm, "memcpy pageable -> pinned, %d thread(s) (spawned)", nt);
        rep(nm, 10, [&] { par_copy(pinned, pageable, N, nt); });
    }
    for (int nt : {1, 4, 8}) {
        char nm[64]; snprintf(nm, sizeof nm, "memcpy pinned -> pageable, %d thread(s) (spawned)", nt);
        rep(nm, 10, [&] { par_copy(pageable2, pinned, N, nt); });
    }
    {   // chunked pipeline: stage chunk i with the CPU while chunk i-1 is on the wire
        const int chunks = 8; const size_t cs = N / chunks;
        rep("pipelined H2D: 8 chunks, memcpy (1 thread) + async DMA", 10, [&] {
            for (int i = 0; i < chunks; ++i) { memcpy(pinned + i * cs, pageable + i * cs, cs); CHECK(hipMemcpyAsync(dev + i * cs, pinned + i * cs, cs, hipMemcpyHostToDevice, s)); }
            CHECK(hipStreamSynchronize(s)); });
        rep("pipelined H2D: 8 chunks, memcpy (4 threads) + async DMA", 10, [&] {
            for (int i = 0; i < chunks; ++i) { par_copy(pinned + i * cs, pageable + i * cs, cs, 4); CHECK(hipMemcpyAsync(dev + i * cs, pinned + i * cs, cs, hipMemcpyHostToDevice, s)); }
            CHECK(hipStreamSynchronize(s)); });
    }
    printf("hardware_concurrency %u\n", std::thread::hardware_concurrency());
    return 0;
}
