// Where do the pinned windows and the packer threads have to sit?  On the two-socket GPU boxes (2 x EPYC, 2 NUMA nodes, the
// process may run on all 256 CPUs but only 16 CPUs' worth: cgroup quota) the host-buffer entry points stage at 22-53 GB/s
// depending on the box.  Measures, for node in {GPU's node, the other}:
//   (a) H2D from a pinned window whose pages live on `node` (hipHostMalloc default, and mmap + mbind + hipHostRegister)
//   (b) N threads pinned to the CPUs of node X copying pageable memory (on node Y) into that window
//   (c) one and two H2D streams
// usage: numa_h2d_probe [MiB]
#include <hip/hip_runtime.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/syscall.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>
#include <thread>
#include <vector>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static const int MPOL_BIND_ = 2;

static std::vector<int> node_cpus(int node) {
    std::ifstream f("/sys/devices/system/node/node" + std::to_string(node) + "/cpulist");
    std::string s;
    std::getline(f, s);
    std::vector<int> out;
    size_t p = 0;
    while (p < s.size()) {
        size_t q = s.find(',', p);
        std::string part = s.substr(p, q == std::string::npos ? std::string::npos : q - p);
        size_t d = part.find('-');
        int a = atoi(part.c_str()), b = d == std::string::npos ? a : atoi(part.c_str() + d + 1);
        for (int c = a; c <= b; ++c) out.push_back(c);
        if (q == std::string::npos) break;
        p = q + 1;
    }
    return out;
}
static void pin_to(const std::vector<int> &cpus) {
    cpu_set_t set;
    CPU_ZERO(&set);
    for (int c : cpus) CPU_SET(c, &set);
    sched_setaffinity(0, sizeof(set), &set);
}
static void *alloc_on_node(size_t bytes, int node) {
    void *p = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    if (p == MAP_FAILED) return nullptr;
    unsigned long mask = 1ul << node;
    if (syscall(SYS_mbind, p, bytes, MPOL_BIND_, &mask, sizeof(mask) * 8, 0) != 0) perror("mbind");
    memset(p, 1, bytes);
    return p;
}
static void par_copy(uint8_t *dst, const uint8_t *src, size_t len, unsigned nthr, const std::vector<int> &cpus) {
    const size_t piece = 1u << 20;
    std::atomic<size_t> next{0};
    auto work = [&]() {
        if (!cpus.empty()) pin_to(cpus);
        for (size_t o; (o = next.fetch_add(piece)) < len;) memcpy(dst + o, src + o, std::min(piece, len - o));
    };
    std::vector<std::thread> th;
    for (unsigned t = 0; t < nthr; ++t) th.emplace_back(work);
    for (auto &t : th) t.join();
}

int main(int argc, char **argv) {
    const size_t total = (size_t)(argc > 1 ? atoi(argv[1]) : 1024) << 20, win = 64u << 20;
    char bdf[64] = {0};
    hipDeviceGetPCIBusId(bdf, sizeof bdf, 0);
    for (char *c = bdf; *c; ++c) *c = (char)tolower(*c);
    int gnode = -1;
    {
        std::ifstream f(std::string("/sys/bus/pci/devices/") + bdf + "/numa_node");
        f >> gnode;
    }
    printf("GPU 0 at %s, NUMA node %d\n", bdf, gnode);
    if (gnode < 0) gnode = 0;
    uint8_t *dev;
    hipMalloc((void **)&dev, 2 * win);
    hipStream_t st[2];
    hipStreamCreateWithFlags(&st[0], hipStreamNonBlocking);
    hipStreamCreateWithFlags(&st[1], hipStreamNonBlocking);
    auto h2d = [&](uint8_t *pin, int nstreams) {
        double best = 0;
        for (int rep = 0; rep < 3; ++rep) {
            double t0 = now();
            int i = 0;
            for (size_t o = 0; o < total; o += win / 2, ++i)
                hipMemcpyAsync(dev + (i & 1) * win, pin + (i & 1) * (win / 2), win / 2, hipMemcpyHostToDevice, st[nstreams == 2 ? (i & 1) : 0]);
            hipStreamSynchronize(st[0]);
            hipStreamSynchronize(st[1]);
            best = std::max(best, total / (now() - t0) / 1e9);
        }
        return best;
    };
    {   // where does hipHostMalloc put its pages?
        uint8_t *pin;
        hipHostMalloc((void **)&pin, win, hipHostMallocDefault);
        memset(pin, 1, win);
        int status[4] = {-9, -9, -9, -9};
        void *pages[4] = {pin, pin + (win / 4), pin + (win / 2), pin + win - 4096};
        syscall(SYS_move_pages, 0, 4ul, pages, nullptr, status, 0);
        printf("hipHostMalloc(default) pages on nodes %d %d %d %d; H2D 1 stream %.1f GB/s, 2 streams %.1f GB/s\n", status[0], status[1],
               status[2], status[3], h2d(pin, 1), h2d(pin, 2));
        hipHostFree(pin);
    }
    for (int node = 0; node < 2; ++node) {
        uint8_t *pin = (uint8_t *)alloc_on_node(win, node);
        if (!pin || hipHostRegister(pin, win, hipHostRegisterDefault) != hipSuccess) {
            printf("node %d: cannot allocate / register\n", node);
            continue;
        }
        printf("pinned window on node %d%s: H2D 1 stream %.1f GB/s, 2 streams %.1f GB/s\n", node, node == gnode ? " (GPU's node)" : "",
               h2d(pin, 1), h2d(pin, 2));
        for (int snode = 0; snode < 2; ++snode) {
            uint8_t *src = (uint8_t *)alloc_on_node(total, snode);
            for (int cnode = 0; cnode < 2; ++cnode) {
                const std::vector<int> cpus = node_cpus(cnode);
                for (unsigned nthr : {8u, 16u}) {
                    double t0 = now();
                    for (size_t o = 0; o < total; o += win) par_copy(pin, src + o, win, nthr, cpus);
                    printf("   copy pageable (node %d) -> window (node %d) with %2u threads on node %d's CPUs: %.1f GB/s\n", snode, node, nthr,
                           cnode, total / (now() - t0) / 1e9);
                }
            }
            munmap(src, total);
        }
        hipHostUnregister(pin);
        munmap(pin, win);
    }
    return 0;
}
