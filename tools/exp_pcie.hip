// exp_pcie.hip -- do uploads and downloads overlap on this box?  (zafx_run_host: chunks over two streams)
//   hipcc -O3 --offload-arch=gfx950 tools/exp_pcie.hip -o tools/bin/exp_pcie ; tools/bin/exp_pcie
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main() {
    const size_t up = 256ull << 20, down = 1024ull << 20;
    CK(hipSetDevice(0));
    for (int flavour = 0; flavour < 3; ++flavour) {
        const unsigned flags = flavour == 0 ? hipHostMallocDefault : flavour == 1 ? hipHostMallocNonCoherent : (hipHostMallocNumaUser | hipHostMallocDefault);
        void *h_up, *h_down, *d_up, *d_down;
        CK(hipHostMalloc(&h_up, up, flags));
        CK(hipHostMalloc(&h_down, down, flags));
        std::memset(h_up, 1, up);
        std::memset(h_down, 1, down);
        CK(hipMalloc(&d_up, up));
        CK(hipMalloc(&d_down, down));
        hipStream_t s0, s1;
        CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking));
        CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
        auto t = [&](auto fn) {
            fn();
            CK(hipDeviceSynchronize());
            double best = 1e9;
            for (int r = 0; r < 3; ++r) {
                const double t0 = now();
                fn();
                CK(hipStreamSynchronize(s0));
                CK(hipStreamSynchronize(s1));
                best = std::min(best, now() - t0);
            }
            return best;
        };
        const double t_up = t([&] { CK(hipMemcpyAsync(d_up, h_up, up, hipMemcpyHostToDevice, s0)); });
        const double t_dn = t([&] { CK(hipMemcpyAsync(h_down, d_down, down, hipMemcpyDeviceToHost, s1)); });
        const double t_both = t([&] {
            CK(hipMemcpyAsync(d_up, h_up, up, hipMemcpyHostToDevice, s0));
            CK(hipMemcpyAsync(h_down, d_down, down, hipMemcpyDeviceToHost, s1));
        });
        // the same from two host threads
        const double t_thr = t([&] {
            std::thread a([&] { (void)hipSetDevice(0); (void)hipMemcpyAsync(d_up, h_up, up, hipMemcpyHostToDevice, s0); (void)hipStreamSynchronize(s0); });
            std::thread b([&] { (void)hipSetDevice(0); (void)hipMemcpyAsync(h_down, d_down, down, hipMemcpyDeviceToHost, s1); (void)hipStreamSynchronize(s1); });
            a.join();
            b.join();
        });
        // download split into 4 pieces on one stream while 4 uploads run on the other
        const double t_pieces = t([&] {
            for (int i = 0; i < 4; ++i) {
                CK(hipMemcpyAsync((char*)d_up + i * (up / 4), (char*)h_up + i * (up / 4), up / 4, hipMemcpyHostToDevice, s0));
                CK(hipMemcpyAsync((char*)h_down + i * (down / 4), (char*)d_down + i * (down / 4), down / 4, hipMemcpyDeviceToHost, s1));
            }
        });
        std::printf("host memory flavour %d: up %.1f GB/s (%.2f ms)  down %.1f GB/s (%.2f ms)  both: %.2f ms (sum %.2f, max %.2f)  two threads: %.2f ms  pieces: %.2f ms\n",
                    flavour, up / t_up / 1e9, t_up * 1e3, down / t_dn / 1e9, t_dn * 1e3, t_both * 1e3, (t_up + t_dn) * 1e3, std::max(t_up, t_dn) * 1e3,
                    t_thr * 1e3, t_pieces * 1e3);
        CK(hipHostFree(h_up));
        CK(hipHostFree(h_down));
        CK(hipFree(d_up));
        CK(hipFree(d_down));
    }
    return 0;
}
