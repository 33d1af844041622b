// device properties the launch heuristics rely on
#include <hip/hip_runtime.h>
#include <cstdio>
int main()
{
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, 0) != hipSuccess) { printf("no device\n"); return 1; }
    printf("name %s arch %s\n", p.name, p.gcnArchName);
    printf("multiProcessorCount %d\n", p.multiProcessorCount);
    printf("maxSharedMemoryPerMultiProcessor %zu\n", (size_t)p.maxSharedMemoryPerMultiProcessor);
    printf("sharedMemPerBlock %zu sharedMemPerBlockOptin %zu\n", (size_t)p.sharedMemPerBlock, (size_t)p.sharedMemPerBlockOptin);
    printf("regsPerBlock %d regsPerMultiprocessor %d warpSize %d\n", p.regsPerBlock, p.regsPerMultiprocessor, p.warpSize);
    printf("clockRate %d kHz memoryClockRate %d kHz memoryBusWidth %d l2CacheSize %d\n", p.clockRate, p.memoryClockRate, p.memoryBusWidth, p.l2CacheSize);
    printf("totalGlobalMem %.1f GB\n", p.totalGlobalMem / 1e9);
    return 0;
}
