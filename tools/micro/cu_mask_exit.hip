// Which sequence around a CU-masked stream keeps a process from finishing?  usage: cu_mask_exit <letters>
//   p: first create, use and destroy two non-blocking streams (before the masked one exists)
//   a: also create a non-blocking stream and launch on it     e: record an event on the masked stream, wait for it on the other
//   d: destroy the masked stream    s: hipDeviceSynchronize after the destroy    m: hipMalloc/hipFree after the destroy
//   c: hipMemsetAsync, device-to-device and device-to-host hipMemcpyAsync on the masked stream
//   2: create, use and destroy a second masked stream afterwards
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) printf("%s: %s\n", #x, hipGetErrorString(e_)); } while (0)
__global__ void k(int* p) { p[threadIdx.x + blockIdx.x * blockDim.x] = 1; }
int main(int argc, char** argv) {
    const char* m = argc > 1 ? argv[1] : "";
    auto has = [&](char c) { return strchr(m, c) != nullptr; };
    uint32_t mask[8] = {0xFFFFFFFFu, 0, 0, 0, 0, 0, 0, 0};
    hipStream_t s, a = nullptr;
    if (has('p')) {
        hipStream_t u, v;
        int* w;
        CK(hipMalloc(&w, 1 << 20));
        CK(hipStreamCreateWithFlags(&u, hipStreamNonBlocking));
        CK(hipStreamCreateWithFlags(&v, hipStreamNonBlocking));
        hipLaunchKernelGGL(k, dim3(256), dim3(256), 0, u, w);
        hipLaunchKernelGGL(k, dim3(256), dim3(256), 0, v, w);
        CK(hipDeviceSynchronize());
        CK(hipStreamDestroy(u));
        CK(hipStreamDestroy(v));
        CK(hipFree(w));
        printf("earlier streams done\n"); fflush(stdout);
    }
    CK(hipExtStreamCreateWithCUMask(&s, 8, mask));
    if (has('a')) CK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking));
    int* p;
    CK(hipMalloc(&p, 1 << 20));
    hipLaunchKernelGGL(k, dim3(256), dim3(256), 0, s, p);
    if (a) hipLaunchKernelGGL(k, dim3(256), dim3(256), 0, a, p);
    if (has('e') && a) {
        hipEvent_t ev;
        CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        CK(hipEventRecord(ev, s));
        CK(hipStreamWaitEvent(a, ev, 0));
        hipLaunchKernelGGL(k, dim3(256), dim3(256), 0, a, p);
        CK(hipStreamSynchronize(a));
        CK(hipEventDestroy(ev));
    }
    if (has('c')) {
        int *q, *h;
        CK(hipMalloc(&q, 1 << 20));
        CK(hipHostMalloc(&h, 1 << 20));
        CK(hipMemsetAsync(q, 0, 1 << 20, s));
        CK(hipMemcpyAsync(q, p, 1 << 20, hipMemcpyDeviceToDevice, s));
        CK(hipMemcpyAsync(h, q, 1 << 20, hipMemcpyDeviceToHost, s));
        CK(hipStreamSynchronize(s));
        int stack_buf[64];
        CK(hipMemcpyAsync(stack_buf, q, sizeof(stack_buf), hipMemcpyDeviceToHost, s));    // pageable destination
        CK(hipStreamSynchronize(s));
        CK(hipHostFree(h));
        CK(hipFree(q));
        printf("copies on the masked stream done (%d)\n", stack_buf[0]); fflush(stdout);
    }
    CK(hipStreamSynchronize(s));
    CK(hipDeviceSynchronize());
    printf("used\n"); fflush(stdout);
    if (has('d')) { CK(hipStreamDestroy(s)); printf("destroyed\n"); fflush(stdout); }
    if (has('s')) { CK(hipDeviceSynchronize()); printf("device sync after destroy\n"); fflush(stdout); }
    if (has('m')) { int* q; CK(hipMalloc(&q, 1 << 24)); CK(hipMemset(q, 0, 1 << 24)); CK(hipFree(q)); printf("malloc after destroy\n"); fflush(stdout); }
    if (has('2')) {
        hipStream_t t;
        CK(hipExtStreamCreateWithCUMask(&t, 8, mask));
        hipLaunchKernelGGL(k, dim3(256), dim3(256), 0, t, p);
        CK(hipStreamSynchronize(t));
        CK(hipStreamDestroy(t));
        printf("second masked stream done\n"); fflush(stdout);
    }
    if (a) { hipLaunchKernelGGL(k, dim3(256), dim3(256), 0, a, p); CK(hipStreamSynchronize(a)); CK(hipStreamDestroy(a)); }
    CK(hipFree(p));
    printf("leaving main (%s)\n", m);
    fflush(stdout);
    return 0;
}
