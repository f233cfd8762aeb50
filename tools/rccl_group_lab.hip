// rccl_group_lab.hip -- what does ONE grouped enqueue over G communicators cost the HOST?  (round 6, VERDICT r05 task 7)
//
// pcx_shard's RCCL transport queues a pass's halo exchange as  ncclGroupStart; G x (ncclSend to the right, ncclRecv from the left);
// ncclGroupEnd  on the G communicators of ncclCommInitAll, one per device.  With one GPU that group cannot be run: a communicator per
// device needs G devices.  What one GPU CAN time is the host side of such a group: G one-rank communicators on device 0 (each its own
// world), every one sending the 2,032-byte halo to itself on its own stream, all inside one group -- the same number of communicators,
// operations and streams the eight-device pass queues, without the wire.
//   hipcc --offload-arch=gfx950 -O2 tools/rccl_group_lab.hip -o tools/rccl_group_lab -lrccl ; tools/rccl_group_lab
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
#define NCCL(x) do { ncclResult_t r_ = (x); if (r_ != ncclSuccess) { fprintf(stderr, "%s: %s\n", #x, ncclGetErrorString(r_)); exit(1); } } while (0)

int main(int argc, char **argv)
{
    const int maxG = 8;
    const size_t halo = 254 * 8;
    HIP(hipSetDevice(0));
    std::vector<ncclComm_t> comm(maxG);
    std::vector<hipStream_t> st(maxG);
    std::vector<char *> a(maxG), b(maxG);
    for (int g = 0; g < maxG; g++) {
        ncclUniqueId id;
        NCCL(ncclGetUniqueId(&id));
        NCCL(ncclCommInitRank(&comm[g], 1, id, 0));
        HIP(hipStreamCreateWithFlags(&st[g], hipStreamNonBlocking));
        HIP(hipMalloc(&a[g], halo));
        HIP(hipMalloc(&b[g], halo));
    }
    const int reps = argc > 1 ? atoi(argv[1]) : 2000;
    for (int G : {1, 2, 4, 8}) {
        auto group = [&]() {
            NCCL(ncclGroupStart());
            for (int g = 0; g < G; g++) {
                NCCL(ncclSend(a[g], halo, ncclChar, 0, comm[g], st[g]));
                NCCL(ncclRecv(b[g], halo, ncclChar, 0, comm[g], st[g]));
            }
            NCCL(ncclGroupEnd());
        };
        for (int i = 0; i < 20; i++) group();                       // connections come up inside the first groups
        for (int g = 0; g < G; g++) HIP(hipStreamSynchronize(st[g]));
        double host = 0;
        const auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < reps; i++) {
            const auto h0 = std::chrono::steady_clock::now();
            group();
            host += std::chrono::duration<double>(std::chrono::steady_clock::now() - h0).count();
            if (i % 16 == 15) for (int g = 0; g < G; g++) HIP(hipStreamSynchronize(st[g]));     // (as a pass's launches would pace it)
        }
        for (int g = 0; g < G; g++) HIP(hipStreamSynchronize(st[g]));
        const double wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        printf("G = %d communicators: %.1f us of host time per grouped enqueue (%d send + %d recv of %zu bytes), %.1f us per group wall incl. the syncs\n",
               G, host / reps * 1e6, G, G, halo, wall / reps * 1e6);
    }
    return 0;
}
