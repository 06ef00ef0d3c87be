// How long it takes the host to learn a count the device has just computed: (A) a 4-byte device-to-host copy into pinned
// memory followed by hipStreamSynchronize, (B) hipStreamSynchronize alone, (C) the kernel itself stores the value and a
// sequence number into host-mapped pinned memory and the host polls the sequence number.  Measurement aid (run on the GPU box).
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ void produce(unsigned *dev, unsigned v) { dev[0] = v; }
__global__ void produce_mapped(unsigned *dev, volatile unsigned *host, unsigned v) {
    dev[0] = v;
    host[0] = v;
    __threadfence_system();
    host[16] = v;            // sequence number, its own cache line
}
__global__ void consume(unsigned *dev, unsigned *sink) { sink[threadIdx.x] = dev[0] + threadIdx.x; }

int main() {
    hipStream_t st; CHK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    unsigned *dev, *sink, *pin; volatile unsigned *mapped; unsigned *mapped_dev;
    CHK(hipMalloc(&dev, 256)); CHK(hipMalloc(&sink, 4096));
    CHK(hipHostMalloc(&pin, 256, hipHostMallocDefault));
    CHK(hipHostMalloc((void **)&mapped, 256, hipHostMallocMapped));
    CHK(hipHostGetDevicePointer((void **)&mapped_dev, (void *)mapped, 0));
    mapped[0] = 0; mapped[16] = 0;
    const int N = 2000;
    for (int mode = 0; mode < 3; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            auto t0 = std::chrono::steady_clock::now();
            unsigned long long sum = 0;
            for (int i = 1; i <= N; ++i) {
                const unsigned v = (unsigned)(mode * 100000 + rep * N + i);
                if (mode == 0) {
                    produce<<<1, 64, 0, st>>>(dev, v);
                    CHK(hipMemcpyAsync(pin, dev, 4, hipMemcpyDeviceToHost, st));
                    CHK(hipStreamSynchronize(st));
                    sum += pin[0];
                } else if (mode == 1) {
                    produce<<<1, 64, 0, st>>>(dev, v);
                    CHK(hipStreamSynchronize(st));
                } else {
                    produce_mapped<<<1, 64, 0, st>>>(dev, mapped_dev, v);
                    while (mapped[16] != v) { }
                    sum += mapped[0];
                }
                consume<<<1, 64, 0, st>>>(dev, sink);       // what follows depends on the count's value only for its launch size
            }
            CHK(hipStreamSynchronize(st));
            const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / N;
            if (rep) printf("%-60s %7.1f us per round trip (kernel, read-back, next kernel)  check %llu\n",
                            mode == 0 ? "4-byte copy to pinned memory + hipStreamSynchronize" : mode == 1 ? "hipStreamSynchronize alone" : "store to host-mapped memory, host polls a sequence number", us, sum % 1000);
        }
    }
    return 0;
}
