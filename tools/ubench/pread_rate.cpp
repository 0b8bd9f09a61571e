// pread_rate — how fast N threads copy a file from the page cache (/dev/shm) into plain and into pinned memory, in chunks of C bytes handed out
// by a counter (round 6: the bedgraph readers deliver 11-12 GB/s, the FASTA readers 31 GB/s from the same file system)
//   hipcc -O2 -o tools/ubench/pread_rate tools/ubench/pread_rate.cpp -lpthread;  tools/ubench/pread_rate FILE
#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <fcntl.h>
#include <cstring>
#include <sys/mman.h>
#include <sys/stat.h>
#include <thread>
#include <unistd.h>
#include <vector>

static double run(int fd, long size, char *dst, long dst_bytes, int n_thr, long chunk)
{
    std::atomic<long> next{0};
    const long n_chunk = (size + chunk - 1) / chunk;
    auto t0 = std::chrono::steady_clock::now();
    std::vector<std::thread> th;
    for (int t = 0; t < n_thr; ++t)
        th.emplace_back([&] {
            for (;;) {
                const long i = next.fetch_add(1);
                if (i >= n_chunk) return;
                const long off = i * chunk, want = size - off < chunk ? size - off : chunk;
                char *d = dst + (off % dst_bytes);          // a ring of dst_bytes
                long got = 0;
                while (got < want) {
                    const ssize_t r = pread(fd, d + got, (size_t)(want - got), off + got);
                    if (r <= 0) return;
                    got += r;
                }
            }
        });
    for (auto &t : th) t.join();
    const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    return size / s / 1e9;
}

// the CLI's way until round 6: per round of `round` bytes, n fresh threads, one part each, joined
static double run_fresh(int fd, long size, char *dst, long dst_bytes, int n_thr, long round)
{
    auto t0 = std::chrono::steady_clock::now();
    for (long base = 0; base < size; base += round) {
        const long want = size - base < round ? size - base : round, part = (want + n_thr - 1) / n_thr;
        std::vector<std::thread> th;
        for (int t = 0; t < n_thr; ++t)
            th.emplace_back([&, t] {
                const long off = base + t * part, w = off >= base + want ? 0 : (base + want - off < part ? base + want - off : part);
                char *d = dst + (off % dst_bytes);
                long got = 0;
                while (got < w) {
                    const ssize_t r = pread(fd, d + got, (size_t)(w - got), off + got);
                    if (r <= 0) return;
                    got += r;
                }
            });
        for (auto &t : th) t.join();
    }
    const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    return size / s / 1e9;
}

// the same copy out of a shared mapping of the file instead of pread(): no mark_page_accessed() per page on the first pass over a freshly written file
static double run_mmap(const char *map, long size, char *dst, long dst_bytes, int n_thr, long chunk)
{
    std::atomic<long> next{0};
    const long n_chunk = (size + chunk - 1) / chunk;
    auto t0 = std::chrono::steady_clock::now();
    std::vector<std::thread> th;
    for (int t = 0; t < n_thr; ++t)
        th.emplace_back([&] {
            for (;;) {
                const long i = next.fetch_add(1);
                if (i >= n_chunk) return;
                const long off = i * chunk, want = size - off < chunk ? size - off : chunk;
                memcpy(dst + (off % dst_bytes), map + off, (size_t)want);
            }
        });
    for (auto &t : th) t.join();
    const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    return size / s / 1e9;
}

int main(int argc, char **argv)
{
    if (argc < 2) return 1;
    const int fd = open(argv[1], O_RDONLY);
    struct stat st;
    if (fd < 0 || fstat(fd, &st)) return 1;
    const long size = st.st_size;
    if (argc > 2) {
        // "quick": what a short-lived process sees — pinned ring of 128 MB, eight threads, the first gigabyte right away, then again and again
        char *pin = nullptr;
        if (hipHostMalloc((void **)&pin, 128L << 20, hipHostMallocDefault) != hipSuccess) return 1;
        const long one = size < (1L << 30) ? size : (1L << 30);
        if (argv[2][0] == 'm') {
            const char *map = (const char *)mmap(nullptr, (size_t)size, PROT_READ, MAP_SHARED, fd, 0);
            if (map == MAP_FAILED) return 1;
            for (int rep = 0; rep < 3; ++rep) printf("quick mmap: pass %d over the first %ld MB, 8 threads, memcpy out of a shared mapping into a pinned ring: %.1f GB/s\n", rep, one >> 20, run_mmap(map, one, pin, 128L << 20, 8, 8L << 20));
            // and pread() over the second gigabyte, which nobody has read yet
            if (size >= (2L << 30)) {
                const int fd2 = fd;
                std::atomic<long> dummy{0};
                (void)dummy;
                // (run() reads [0, one): shift by reading through an offset file descriptor is not possible with pread: use a second mapping-free pass over bytes [1 GB, 2 GB) by hand)
                auto t0 = std::chrono::steady_clock::now();
                std::vector<std::thread> th;
                std::atomic<long> next{0};
                for (int t = 0; t < 8; ++t)
                    th.emplace_back([&] {
                        for (;;) {
                            const long i = next.fetch_add(1);
                            if (i >= 128) return;
                            const long off = (1L << 30) + i * (8L << 20);
                            long got = 0;
                            while (got < (8L << 20)) {
                                const ssize_t r = pread(fd2, pin + (off % (128L << 20)) + got, (size_t)((8L << 20) - got), off + got);
                                if (r <= 0) return;
                                got += r;
                            }
                        }
                    });
                for (auto &t : th) t.join();
                printf("quick mmap: then pread() over the SECOND gigabyte (first read of those pages): %.1f GB/s\n", (1L << 30) / std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() / 1e9);
            }
            return 0;
        }
        for (int rep = 0; rep < 6; ++rep) printf("quick: pass %d over the first %ld MB, 8 threads, chunks of 8 MB into a pinned ring: %.1f GB/s\n", rep, one >> 20, run(fd, one, pin, 128L << 20, 8, 8L << 20));
        return 0;
    }
    const long ring = 512L << 20;
    char *plain = (char *)malloc(ring), *pinned = nullptr;
    if (hipHostMalloc((void **)&pinned, ring, hipHostMallocDefault) != hipSuccess) pinned = nullptr;
    for (long i = 0; i < ring; i += 4096) plain[i] = 1;
    printf("file %ld bytes; ring %ld MB\n", size, ring >> 20);
    for (long chunk : {8L << 20, 64L << 20})
        for (int n : {1, 4, 8, 16, 32})
            printf("chunk %3ld MB, %2d threads: plain %.1f GB/s, pinned %.1f GB/s, pinned ring of 128 MB %.1f GB/s\n", chunk >> 20, n, run(fd, size, plain, ring, n, chunk),
                   pinned ? run(fd, size, pinned, ring, n, chunk) : 0.0, pinned ? run(fd, size, pinned, 128L << 20, n, chunk) : 0.0);
    // the same while another thread keeps host-to-device copies of 64 MB from a second pinned buffer in flight (what the CLI's uploads do beside its readers)
    if (pinned) {
        char *src2 = nullptr, *dev = nullptr;
        (void)hipHostMalloc((void **)&src2, 64L << 20, hipHostMallocDefault);
        (void)hipMalloc((void **)&dev, 64L << 20);
        hipStream_t st;
        (void)hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
        std::atomic<bool> stop{false};
        std::atomic<long> copied{0};
        std::thread dma([&] {
            while (!stop) {
                (void)hipMemcpyAsync(dev, src2, 64L << 20, hipMemcpyHostToDevice, st);
                (void)hipStreamSynchronize(st);
                copied += 64L << 20;
            }
        });
        for (int n : {8, 16}) {
            const long c0 = copied;
            auto t0 = std::chrono::steady_clock::now();
            const double r = run(fd, size, pinned, ring, n, 8L << 20);
            const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            printf("chunk 8 MB, %2d threads, pinned, BESIDE host-to-device copies (%.1f GB/s of them): %.1f GB/s\n", n, (copied - c0) / s / 1e9, r);
        }
        stop = true;
        dma.join();
    }
    for (int n : {4, 8, 16})
        printf("rounds of 64 MB, %2d FRESH threads per round: plain %.1f GB/s, pinned %.1f GB/s\n", n, run_fresh(fd, size, plain, ring, n, 64L << 20), pinned ? run_fresh(fd, size, pinned, ring, n, 64L << 20) : 0.0);
    return 0;
}
