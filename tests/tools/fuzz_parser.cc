// Parser fuzz (development tool): mutated .evc streams (bit flips, overwritten bytes, truncations, garbage runs) through xhost_parser, for sanitizer builds:
//   g++ -O1 -g -std=c++17 -fsanitize=address,undefined -fno-sanitize=shift-base -pthread -o fuzz tests/tools/fuzz_parser.cc xevd_amd/host/evc_parser.cc xevd_amd/host/evc_writer.cc xevd_amd/host/xwq.cc
//   g++ -O1 -g -std=c++17 -fsanitize=thread -pthread -o fuzz_tsan ...        (tile streams with threads > 1: the tiles of a picture parse in parallel)
//   ./fuzz <mutations per stream> <parser threads> a.evc b.evc ...             (golden streams: np.load(tests/golden/stream_*.npz)["bytes"])
// Round 2: 8250 mutations of the 55 golden streams under ASan + UBSan and the tiled ones under TSan with 4 threads - clean after the tile test was moved in
// front of every neighbour read (another tile's maps may be written at that moment).
// Round 4 (persistent parser + rebind, windowed bit reader, coefficients decoded in place, pooled tile threads): 120 mutations of each of the 51 golden streams under ASan + UBSan,
// 40 of each tiled one under TSan with 4 threads - see DESIGN 5b.
// Repeated after sps_suco_flag went in (right-hand neighbour reads in every derivation): 1500 mutations of each of the five SUCO golden streams under ASan + UBSan, 300 of the tiled one under TSan with 4 threads - clean.
// Repeated after BTT and the local dual tree went in: 1200 + 1200 mutations of their six golden streams (ASan + UBSan), 300 of the tiled ones under TSan with 4 threads - clean.
#include "../../include/xevd_host.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <string>
static uint64_t rng = 88172645463325252ull;
static uint32_t rnd() { rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17; return (uint32_t)(rng >> 11); }
int main(int argc, char **argv)
{
    int iters = atoi(argv[1]), threads = atoi(argv[2]);
    long pics = 0, errs = 0;
    // round 4: ONE parser object for the whole run, rebound to every mutated stream (xhost_parser_rebind: what a work-queue worker does from GOP to GOP) - a parser that
    // has just failed in the middle of a picture, or decoded a stream of another geometry / tool set, must behave like a new one; every fourth stream a new parser
    xhost_parser *p = nullptr;
    long n_streams = 0;
    for (int f = 3; f < argc; f++) {
        FILE *fp = fopen(argv[f], "rb"); if (!fp) continue;
        std::vector<uint8_t> d; uint8_t buf[65536]; size_t n;
        while ((n = fread(buf, 1, sizeof buf, fp)) > 0) d.insert(d.end(), buf, buf + n);
        fclose(fp);
        for (int it = 0; it < iters; it++) {
            std::vector<uint8_t> m = d;
            const int kind = rnd() % 4, nm = 1 + rnd() % 3;
            for (int k = 0; k < nm; k++) {
                const size_t pos = rnd() % m.size();
                if (kind == 0) m[pos] ^= (uint8_t)(1u << (rnd() % 8));
                else if (kind == 1) m[pos] = (uint8_t)rnd();
                else if (kind == 2) { m.resize(pos + 1); break; }
                else { const size_t len = 1 + rnd() % 8; for (size_t q = pos; q < pos + len && q < m.size(); q++) m[q] = (uint8_t)rnd(); }
            }
            if (p && (n_streams++ & 3) == 3) { xhost_parser_close(p); p = nullptr; }
            if (p) { if (xhost_parser_rebind(p, m.data(), m.size()) < 0) return 1; }
            else { p = xhost_parser_open(m.data(), m.size()); if (threads > 1) xhost_parser_set_threads(p, threads); }
            xhost_picture pic;
            for (;;) {
                const int rc = xhost_parser_next(p, &pic);
                if (rc == 0) break;
                if (rc < 0) { errs++; break; }
                pics++;
                if (pic.n_dmvr_sub) { std::vector<int16_t> mv((size_t)pic.n_dmvr_sub * 4, 0); xhost_parser_set_dmvr_mvs(p, mv.data(), pic.n_dmvr_sub); }
            }
        }
    }
    if (p) xhost_parser_close(p);
    printf("pictures %ld, failed streams %ld\n", pics, errs);
    return 0;
}
