// tools/segv/repro_lib.cpp -- VERDICT r3 item 3: the SIGSEGV inside hipLaunchKernel seen in 4 of 19 runs of bench.py under
// `rocprofv3 --kernel-trace` (worker threads of the transcript batch; 0 of 80 runs without the profiler).
//   repro_lib <mode> <calls>      mode a: lock-step off, 1 worker      b: lock-step off, 16 workers
//                                      c: lock-step on, 1 lane          d: lock-step on, 6 lanes
//                                      e: like b with zkhip_set_fri_graph(0): no hipGraphLaunch anywhere
// Each call is one zkhip_prove_transcripts of 64 transcripts of 13 221 bytes (the bench's batch64).  Plain C++ over the C ABI.
// Prints the calls completed; tools/segv/run.py starts it many times with and without the profiler and counts the exit statuses.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/zkhip.h"
#include "../../include/zkhip_chips.h"

int main(int argc, char** argv) {
    if (argc < 3) { std::fprintf(stderr, "usage: %s a|b|c|d|e <calls>\n", argv[0]); return 1; }
    const char mode = argv[1][0];
    const int calls = std::atoi(argv[2]);
    if (zkhip_device_count() <= 0) { std::fprintf(stderr, "no device\n"); return 2; }
    const zkhip_params prm = ZKHIP_PARAMS_SP1_CORE;
    const int n = 64;
    std::vector<std::vector<uint8_t>> msgs(n), proofs(n);
    std::vector<zkhip_transcript_job> jobs(n);
    for (int i = 0; i < n; i++) {
        msgs[i].resize(13217 + 4);
        for (size_t k = 0; k < msgs[i].size(); k++) msgs[i][k] = (uint8_t)(k * 131 + i * 7);
        const size_t cap = zkhip_sha256_machine_proof_size(msgs[i].size(), &prm);
        proofs[i].resize(cap);
        std::memset(&jobs[i], 0, sizeof jobs[i]);
        jobs[i].message = msgs[i].data(); jobs[i].message_len = msgs[i].size();
        jobs[i].proof = proofs[i].data(); jobs[i].proof_cap = cap;
    }
    int in_flight = 4;
    if (mode == 'a') { zkhip_set_lockstep(0, 0); in_flight = 1; }
    else if (mode == 'b' || mode == 'e') { zkhip_set_lockstep(0, 0); in_flight = 16; if (mode == 'e') zkhip_set_fri_graph(0); }
    else if (mode == 'c') zkhip_set_lockstep(16, 1);
    else zkhip_set_lockstep(16, 6);
    uint32_t vk[8];
    for (int c = 0; c < calls; c++) {
        if (zkhip_prove_transcripts(nullptr, 0, jobs.data(), n, &prm, in_flight, /*verify=*/(c & 1), vk) != ZKHIP_OK) { std::fprintf(stderr, "call %d: %s\n", c, zkhip_last_error()); return 3; }
    }
    std::printf("mode %c: %d calls done, fiber stack high water %llu bytes\n", mode, calls, (unsigned long long)zkhip_lockstep_stack_high_water());
    std::fflush(stdout);
    zkhip_release_cached_contexts();
    return 0;
}
