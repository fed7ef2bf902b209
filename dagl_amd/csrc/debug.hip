// Ablation builds only (-DDAGL_ABLATION): block phase stamps of the matrix-core kernels.  A kernel given a stamp buffer
// writes four 100 MHz time stamps per block (entry, loop start, loop end, exit); with DAGL_TIMES_FILE set the launcher
// synchronises and appends them to that file, one line per launch: "<kernel> <blocks> t0 t1 t2 t3 t0 t1 ...".
// tools/block_times.py turns the lines into a timeline (dispatch ramp, prologue, loop, epilogue, tail).
#include <stdio.h>
#include <stdlib.h>

#include <vector>

#include "dagl_common.h"

namespace dagl {

#ifdef DAGL_ABLATION
unsigned long long* dbg_times_buffer(size_t blocks) {
    static unsigned long long* buf = nullptr;
    static size_t cap = 0;
    if (blocks > cap) {
        if (buf) (void)hipFree(buf);
        if (hipMalloc(&buf, blocks * 4 * sizeof(unsigned long long)) != hipSuccess) { buf = nullptr; cap = 0; return nullptr; }
        cap = blocks;
    }
    return buf;
}

void dbg_times_dump(hipStream_t s, const char* kernel, const unsigned long long* buf, size_t blocks) {
    const char* path = getenv("DAGL_TIMES_FILE");
    if (!path || !buf) return;
    static int skip = getenv("DAGL_TIMES_SKIP") ? atoi(getenv("DAGL_TIMES_SKIP")) : 0;   // dump calls passed over first (warm clocks)
    static int budget = 48;                                   // dump calls recorded per process
    if (skip > 0) { --skip; return; }
    if (budget-- <= 0) return;
    if (hipStreamSynchronize(s) != hipSuccess) return;
    std::vector<unsigned long long> h(blocks * 4);
    if (hipMemcpy(h.data(), buf, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost) != hipSuccess) return;
    FILE* f = fopen(path, "a");
    if (!f) return;
    fprintf(f, "%s %zu", kernel, blocks);
    for (unsigned long long v : h) fprintf(f, " %llu", v);
    fprintf(f, "\n");
    fclose(f);
}
#endif

}  // namespace dagl
