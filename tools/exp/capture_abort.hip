// What is left of a stream capture that is given up half way on this runtime?  (tools/exp: an experiment, not part of the library.)
//   hipcc --offload-arch=gfx950 -o /tmp/capture_abort tools/exp/capture_abort.hip && /tmp/capture_abort
#include <hip/hip_runtime.h>
#include <cstdio>
#include <thread>
__global__ void k_inc(int* p) { atomicAdd(p, 1); }
static const char* st(hipStream_t s) {
    hipStreamCaptureStatus cs; hipError_t e = hipStreamIsCapturing(s, &cs);
    static char buf[4][96]; static int i = 0; char* b = buf[i++ & 3];
    snprintf(b, 96, "%s/%d", e == hipSuccess ? "ok" : hipGetErrorName(e), e == hipSuccess ? (int)cs : -1); (void)hipGetLastError(); return b;
}
static void launch_check(const char* what, hipStream_t s, int* d) {
    hipLaunchKernelGGL(k_inc, dim3(1), dim3(1), 0, s, d);
    hipError_t e = hipGetLastError(); hipError_t e2 = hipStreamSynchronize(s);
    printf("  %-34s launch %s, sync %s\n", what, hipGetErrorName(e), hipGetErrorName(e2)); (void)hipGetLastError();
}
int main() {
    int* d; hipMalloc(&d, 4); hipMemset(d, 0, 4);
    hipStream_t a, b; hipStreamCreateWithFlags(&a, hipStreamNonBlocking); hipStreamCreateWithFlags(&b, hipStreamNonBlocking);
    hipEvent_t ev; hipEventCreateWithFlags(&ev, hipEventDisableTiming);
    hipGraph_t g = nullptr;
    printf("case 1: fork a -> b, never joined, EndCapture(a)\n");
    hipStreamBeginCapture(a, hipStreamCaptureModeRelaxed);
    hipLaunchKernelGGL(k_inc, dim3(1), dim3(1), 0, a, d);
    hipEventRecord(ev, a); hipStreamWaitEvent(b, ev, 0);
    hipLaunchKernelGGL(k_inc, dim3(1), dim3(1), 0, b, d);
    printf("  inside: a %s  b %s\n", st(a), st(b));
    hipError_t e = hipStreamEndCapture(a, &g); printf("  EndCapture(a): %s graph %p\n", hipGetErrorName(e), (void*)g); (void)hipGetLastError();
    printf("  after: a %s  b %s\n", st(a), st(b));
    launch_check("a after", a, d); launch_check("b after", b, d);
    if (g) { hipGraphDestroy(g); g = nullptr; }
    e = hipStreamEndCapture(b, &g); printf("  EndCapture(b): %s\n", hipGetErrorName(e)); (void)hipGetLastError();
    printf("  then: a %s  b %s\n", st(a), st(b));
    launch_check("b after EndCapture(b)", b, d);
    // a fresh pair for the next case if b is lost
    hipStream_t c, f; hipStreamCreateWithFlags(&c, hipStreamNonBlocking); hipStreamCreateWithFlags(&f, hipStreamNonBlocking);
    printf("case 2: hipStreamSynchronize inside the capture, EndCapture\n");
    hipStreamBeginCapture(c, hipStreamCaptureModeRelaxed);
    hipLaunchKernelGGL(k_inc, dim3(1), dim3(1), 0, c, d);
    e = hipStreamSynchronize(c); printf("  sync inside: %s\n", hipGetErrorName(e)); (void)hipGetLastError();
    printf("  inside: c %s\n", st(c));
    e = hipStreamEndCapture(c, &g); printf("  EndCapture(c): %s graph %p\n", hipGetErrorName(e), (void*)g); (void)hipGetLastError();
    printf("  after: c %s\n", st(c));
    launch_check("c after", c, d);
    if (g) { hipGraphDestroy(g); g = nullptr; }
    printf("case 3: an event recorded before the capture, queried from inside it / an event recorded inside, queried after\n");
    hipEvent_t e0, e1; hipEventCreateWithFlags(&e0, hipEventDisableTiming); hipEventCreateWithFlags(&e1, hipEventDisableTiming);
    hipEventRecord(e0, f); hipStreamSynchronize(f);
    hipStreamBeginCapture(f, hipStreamCaptureModeRelaxed);
    hipLaunchKernelGGL(k_inc, dim3(1), dim3(1), 0, f, d);
    e = hipEventQuery(e0); printf("  query(e0 recorded before) inside: %s\n", hipGetErrorName(e)); (void)hipGetLastError();
    hipEventRecord(e1, f);
    e = hipEventQuery(e1); printf("  query(e1 recorded inside) inside: %s\n", hipGetErrorName(e)); (void)hipGetLastError();
    printf("  inside: f %s\n", st(f));
    e = hipStreamEndCapture(f, &g); printf("  EndCapture(f): %s graph %p\n", hipGetErrorName(e), (void*)g); (void)hipGetLastError();
    e = hipEventQuery(e1); printf("  query(e1) after: %s\n", hipGetErrorName(e)); (void)hipGetLastError();
    launch_check("f after", f, d);
    // case 4: ONLY the query of an event recorded on the stream before its capture began, from ANOTHER thread (what a collective
    // library's watchdog does with the end events of earlier collectives): does it break the capture?
    printf("case 4: an event recorded on the stream BEFORE its capture, queried from another thread during the capture\n");
    hipStream_t s4; hipStreamCreateWithFlags(&s4, hipStreamNonBlocking);
    hipEvent_t e4; hipEventCreateWithFlags(&e4, hipEventDisableTiming);
    hipLaunchKernelGGL(k_inc, dim3(1), dim3(1), 0, s4, d); hipEventRecord(e4, s4); hipStreamSynchronize(s4);
    hipStreamBeginCapture(s4, hipStreamCaptureModeRelaxed);
    hipLaunchKernelGGL(k_inc, dim3(1), dim3(1), 0, s4, d);
    { hipError_t qe = hipSuccess; std::thread th([&] { qe = hipEventQuery(e4); (void)hipGetLastError(); }); th.join(); printf("  query from a thread: %s\n", hipGetErrorName(qe)); }
    printf("  inside after the query: s4 %s\n", st(s4));
    e = hipStreamEndCapture(s4, &g); printf("  EndCapture(s4): %s graph %p\n", hipGetErrorName(e), (void*)g); (void)hipGetLastError();
    launch_check("s4 after", s4, d);
    if (g) { hipGraphExec_t ex = nullptr; e = hipGraphInstantiate(&ex, g, nullptr, nullptr, 0); printf("  instantiate: %s\n", hipGetErrorName(e)); if (ex) { e = hipGraphLaunch(ex, s4); hipStreamSynchronize(s4); printf("  launch: %s\n", hipGetErrorName(e)); } g = nullptr; }
    // case 5: the same with an event that was recorded on ANOTHER stream
    printf("case 5: an event of another stream queried from another thread during the capture\n");
    hipStream_t s5, o5; hipStreamCreateWithFlags(&s5, hipStreamNonBlocking); hipStreamCreateWithFlags(&o5, hipStreamNonBlocking);
    hipEvent_t e5; hipEventCreateWithFlags(&e5, hipEventDisableTiming);
    hipLaunchKernelGGL(k_inc, dim3(1), dim3(1), 0, o5, d); hipEventRecord(e5, o5); hipStreamSynchronize(o5);
    hipStreamBeginCapture(s5, hipStreamCaptureModeRelaxed);
    hipLaunchKernelGGL(k_inc, dim3(1), dim3(1), 0, s5, d);
    { hipError_t qe = hipSuccess; std::thread th([&] { qe = hipEventQuery(e5); (void)hipGetLastError(); }); th.join(); printf("  query from a thread: %s\n", hipGetErrorName(qe)); }
    printf("  inside after the query: s5 %s\n", st(s5));
    e = hipStreamEndCapture(s5, &g); printf("  EndCapture(s5): %s graph %p\n", hipGetErrorName(e), (void*)g); (void)hipGetLastError();
    launch_check("s5 after", s5, d);
    // case 6: an event recorded INSIDE a capture that ended well, queried afterwards (a captured collective's end event)
    printf("case 6: an event recorded inside a capture that ended well, queried afterwards\n");
    hipStream_t s6; hipStreamCreateWithFlags(&s6, hipStreamNonBlocking);
    hipEvent_t e6; hipEventCreateWithFlags(&e6, hipEventDisableTiming);
    hipStreamBeginCapture(s6, hipStreamCaptureModeRelaxed);
    hipLaunchKernelGGL(k_inc, dim3(1), dim3(1), 0, s6, d); hipEventRecord(e6, s6);
    g = nullptr; e = hipStreamEndCapture(s6, &g); printf("  EndCapture(s6): %s\n", hipGetErrorName(e)); (void)hipGetLastError();
    e = hipEventQuery(e6); printf("  query after: %s\n", hipGetErrorName(e)); (void)hipGetLastError();
    launch_check("s6 after", s6, d);
    int h = 0; hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost); printf("kernels that ran: %d\n", h);
    return 0;
}
