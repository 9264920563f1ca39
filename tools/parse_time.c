/* parse_time.c - the host front end alone: milliseconds per picture inside xhost_parser_next (no GPU).  usage: parse_time in.evc [threads] [repeats]
 * build: gcc -O2 -I include -o /tmp/parse_time tools/parse_time.c -L xevd_amd -lxevd_host -Wl,-rpath,$PWD/xevd_amd */
#define _XOPEN_SOURCE 700
#include <stdio.h>
#include <stdlib.h>
#include <time.h>
#include "xevd_host.h"
static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }
int main(int argc, char **argv)
{
    if (argc < 2) return 2;
    FILE *f = fopen(argv[1], "rb");
    if (!f) return 1;
    fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET);
    uint8_t *d = (uint8_t *)malloc((size_t)n);
    if (fread(d, 1, (size_t)n, f) != (size_t)n) return 1;
    const int threads = argc > 2 ? atoi(argv[2]) : 1, reps = argc > 3 ? atoi(argv[3]) : 1;
    for (int r = 0; r < reps; r++) {
        xhost_parser *p = xhost_parser_open(d, (size_t)n);
        xhost_parser_set_threads(p, threads);
        xhost_picture pic;
        int k = 0, rc;
        double t0 = now(), tot = 0;
        while ((rc = xhost_parser_next(p, &pic)) == 1) {
            const double t1 = now();
            if (r == reps - 1) printf("picture %2d poc %3d n_cu %7d n_coef %9zu: %.2f ms\n", k, pic.poc, pic.batch.n_cu, pic.batch.n_coef, 1e3 * (t1 - t0));
            tot += t1 - t0; k++; t0 = now();
        }
        if (rc < 0) printf("error %d: %s\n", rc, xhost_parser_error(p));
        printf("pass %d: %d pictures, %.2f ms per picture (%d threads)\n", r, k, 1e3 * tot / (k ? k : 1), threads);
        xhost_parser_close(p);
    }
    return 0;
}
