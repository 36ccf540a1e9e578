// The reference's call shape under load, from C++: T host threads, one detector each (MarkerDetector is not re-entrant: one object per thread,
// /root/reference/src/markerdetector.cpp:334,372-380), each calling arucohip_detect() on its own pinned host frame in a loop - the frame loop of
// /root/reference/utils/aruco_test.cpp:153-160 with T cameras. Prints one JSON object {"T": frames/s over all threads, ...}.
//   thread_bench <gray.raw> <width> <height> <seconds> T [T ...]
// Built by __graft_entry__.build() into build/thread_bench (hipcc, links aruco_amd/libarucohip.so); bench.py's latency leg runs it.
#include <hip/hip_runtime_api.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "../include/arucohip.h"

int main(int argc, char** argv) {
    if (argc < 6) {
        fprintf(stderr, "usage: thread_bench gray.raw width height seconds T [T ...]\n");
        return 2;
    }
    const int W = atoi(argv[2]), H = atoi(argv[3]);
    const double seconds = atof(argv[4]);
    std::vector<uint8_t> frame((size_t)W * H);
    FILE* f = fopen(argv[1], "rb");
    if (!f || fread(frame.data(), 1, frame.size(), f) != frame.size()) {
        fprintf(stderr, "cannot read %s\n", argv[1]);
        return 2;
    }
    fclose(f);
    printf("{");
    for (int a = 5; a < argc; a++) {
        const int T = atoi(argv[a]);
        std::vector<arucohip_handle*> hs(T, nullptr);
        std::vector<uint8_t*> pinned(T, nullptr);
        std::vector<long> counts(T, 0);
        std::vector<int> found(T, 0);
        for (int i = 0; i < T; i++) {
            if (arucohip_create(nullptr, 0, W, H, 1, &hs[i]) != ARUCOHIP_OK) {
                fprintf(stderr, "create failed\n");
                return 1;
            }
            if (hipHostMalloc((void**)&pinned[i], frame.size(), hipHostMallocDefault) != hipSuccess) return 1;
            memcpy(pinned[i], frame.data(), frame.size());
        }
        std::atomic<bool> go{false}, stop{false};
        std::atomic<int> failed{0};
        auto work = [&](int i) {
            std::vector<arucohip_marker_t> out(256);
            int n = 0;
            for (int k = 0; k < 5; k++)   // first calls: eager sizing, graph capture
                if (arucohip_detect(hs[i], pinned[i], W, H, (size_t)W, nullptr, nullptr, 0, -1.0f, 0, out.data(), 256, &n) != ARUCOHIP_OK) failed++;
            while (!go.load()) std::this_thread::yield();
            while (!stop.load()) {
                if (arucohip_detect(hs[i], pinned[i], W, H, (size_t)W, nullptr, nullptr, 0, -1.0f, 0, out.data(), 256, &n) != ARUCOHIP_OK) {
                    failed++;
                    break;
                }
                counts[i]++;
            }
            found[i] = n;
        };
        std::vector<std::thread> th;
        for (int i = 0; i < T; i++) th.emplace_back(work, i);
        std::this_thread::sleep_for(std::chrono::milliseconds(300));   // every thread is past its first calls (they wait for `go` in any case)
        const auto t0 = std::chrono::steady_clock::now();
        go = true;
        std::this_thread::sleep_for(std::chrono::duration<double>(seconds));
        stop = true;
        for (auto& t : th) t.join();
        const double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        long total = 0;
        for (long c : counts) total += c;
        if (failed.load()) {
            fprintf(stderr, "detect failed: %s\n", arucohip_last_error_string(hs[0]));
            return 1;
        }
        printf("%s\"%d\": %.1f", a > 5 ? ", " : "", T, total / el);
        if (a == argc - 1) printf(", \"markers\": %d", found[0]);
        for (int i = 0; i < T; i++) {
            arucohip_destroy(hs[i]);
            (void)hipHostFree(pinned[i]);
        }
    }
    printf("}\n");
    return 0;
}
