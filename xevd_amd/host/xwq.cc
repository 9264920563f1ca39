// xwq.cc - host work queue for multi-GPU decoding (include/xevd_wq.h): a mutex + condition variable FIFO, one worker thread per device.
#include "../../include/xevd_wq.h"
#include "../../include/xevd_host.h"

#include <condition_variable>
#include <cstring>
#include <deque>
#include <mutex>
#include <thread>
#include <vector>

struct xwq {
    std::mutex mu;
    std::condition_variable cv;
    std::deque<xwq_job> jobs;
    bool closed = false;
};

extern "C" {

xwq *xwq_create(void) { return new (std::nothrow) xwq(); }
void xwq_destroy(xwq *q) { delete q; }

int xwq_push(xwq *q, const xwq_job *job)
{
    if (!q || !job) return XWQ_ERR_INVALID_ARGUMENT;
    {
        std::lock_guard<std::mutex> g(q->mu);
        if (q->closed) return XWQ_ERR_UNEXPECTED;
        q->jobs.push_back(*job);
    }
    q->cv.notify_one();
    return 0;
}

void xwq_close(xwq *q)
{
    if (!q) return;
    { std::lock_guard<std::mutex> g(q->mu); q->closed = true; }
    q->cv.notify_all();
}

int xwq_pop(xwq *q, xwq_job *out)
{
    if (!q || !out) return XWQ_ERR_INVALID_ARGUMENT;
    std::unique_lock<std::mutex> g(q->mu);
    q->cv.wait(g, [&] { return !q->jobs.empty() || q->closed; });
    if (q->jobs.empty()) return 0;
    *out = q->jobs.front();
    q->jobs.pop_front();
    return 1;
}

int xwq_run(xwq *q, const int *devices, int n_devices, xwq_init_fn init, xwq_job_fn job, xwq_fini_fn fini, void *user, int *jobs_done)
{
    if (!q || !devices || n_devices <= 0 || !job) return XWQ_ERR_INVALID_ARGUMENT;
    std::vector<std::thread> th;
    std::vector<int> done((size_t)n_devices, 0), rc((size_t)n_devices, 0), down((size_t)n_devices, 0);      // down: the worker never came up (its own flag: a job may fail with any code, -105 included)
    for (int i = 0; i < n_devices; i++)
        th.emplace_back([&, i] {
            void *st = init ? init(devices[i], user) : nullptr;
            if (init && !st) { down[(size_t)i] = 1; return; }                             // this device is out; the others take its share
            xwq_job j;
            while (xwq_pop(q, &j) == 1) {
                const int r = job(st, &j);
                if (r < 0 && rc[(size_t)i] == 0) rc[(size_t)i] = r;
                if (r >= 0) done[(size_t)i]++;
            }
            if (fini) fini(st);
        });
    for (auto &t : th) t.join();
    int first = 0, usable = 0;
    for (int i = 0; i < n_devices; i++) {
        if (jobs_done) jobs_done[i] = done[(size_t)i];
        if (!down[(size_t)i]) usable++;
        if (rc[(size_t)i] < 0 && first == 0) first = rc[(size_t)i];                       // every failed job is reported, whatever its code
    }
    if (!usable) return XWQ_ERR_UNEXPECTED;                                                // no worker came up: nothing was decoded
    return first;
}

// NAL unit type of the 2-byte header (xevd_eco.c:1178-1209): 1 forbidden bit, 6 bits type + 1, 3 bits temporal id, ...
static int nal_type(const uint8_t *p) { return ((((int)p[0] << 8) | p[1]) >> 9 & 63) - 1; }
enum { NUT_NONIDR = 0, NUT_IDR = 1, NUT_SPS = 24, NUT_PPS = 25, NUT_APS = 26 };

int xwq_split_gops(const uint8_t *data, size_t size, int stream, xwq_job *jobs, int max_jobs)
{
    if (!data || !jobs || max_jobs <= 0) return XWQ_ERR_INVALID_ARGUMENT;
    int n = 0, pictures = 0;
    size_t pos = 0;
    xhost_scan *scan = xhost_scan_open();                // picture boundaries: a picture may come as several slice NAL units
    if (!scan) return XWQ_ERR_INVALID_ARGUMENT;
    while (pos + 4 <= size) {
        const size_t len = ((size_t)data[pos] << 24) | ((size_t)data[pos + 1] << 16) | ((size_t)data[pos + 2] << 8) | data[pos + 3];
        if (len < 2 || pos + 4 + len > size) { xhost_scan_close(scan); return -202; }
        const int t = nal_type(data + pos + 4);
        const int kind = xhost_scan_nal(scan, data + pos + 4, len);      // 1: first slice of a picture, 2: a further slice
        if (kind < 0) { xhost_scan_close(scan); return -202; }
        if (t == NUT_IDR && kind == 1) {
            if (n == max_jobs) { xhost_scan_close(scan); return -203; }          // more units than the caller's array holds: nothing is dropped silently - grow and call again
            if (n) jobs[n - 1].size = pos - jobs[n - 1].offset;
            memset(&jobs[n], 0, sizeof(jobs[n]));
            jobs[n].stream = stream; jobs[n].unit = n; jobs[n].offset = pos; jobs[n].first_picture = pictures;
            n++;
        }
        if (kind == 1 && n) { jobs[n - 1].n_pictures++; pictures++; }
        pos += 4 + len;
    }
    xhost_scan_close(scan);
    if (n) jobs[n - 1].size = pos - jobs[n - 1].offset;
    return n;
}

size_t xwq_unit_bytes(const uint8_t *data, size_t size, const xwq_job *job, uint8_t *out, size_t cap)
{
    if (!data || !job || !out || job->offset + job->size > size) return 0;
    size_t pos = 0, w = 0;
    while (pos + 4 <= job->offset) {
        const size_t len = ((size_t)data[pos] << 24) | ((size_t)data[pos + 1] << 16) | ((size_t)data[pos + 2] << 8) | data[pos + 3];
        if (len < 2 || pos + 4 + len > size) return 0;
        const int t = nal_type(data + pos + 4);
        if (t == NUT_SPS || t == NUT_PPS || t == NUT_APS) {
            if (w + 4 + len > cap) return 0;
            memcpy(out + w, data + pos, 4 + len);
            w += 4 + len;
        }
        pos += 4 + len;
    }
    if (w + job->size > cap) return 0;
    memcpy(out + w, data + job->offset, job->size);
    return w + job->size;
}

}   // extern "C"
