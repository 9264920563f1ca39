/*
 * xevd_wq.h - the host work queue that shards independent decoding jobs over the GPUs of one node (libxevd_host.so, plain C ABI).
 *
 * SURVEY 8(e) / north_star: streams and closed GOPs (IDR to IDR) are independent units with no exchange step, so multi-GPU decoding is a
 * plain host queue - one worker thread + one xgpu_ctx per device, each pulling the next job when it is done with the last (dynamic: a slow
 * job or a slow device does not hold the others up), no collective, no RCCL.  What it stands in for in the reference: the thread pool that
 * hands CTU rows / tiles to up to 8 host threads (src_base/xevd_tp.c, xevd.c:1470-1526) - here the unit is a whole GOP and the worker a GPU.
 */
#ifndef XEVD_WQ_H
#define XEVD_WQ_H

#include <stddef.h>
#include <stdint.h>

#define XWQ_ERR_INVALID_ARGUMENT (-101)  /* XEVD_ERR_INVALID_ARGUMENT (inc/xevd.h:61)                                          */
#define XWQ_ERR_UNEXPECTED       (-105)  /* XEVD_ERR_UNEXPECTED (inc/xevd.h:65): push into a closed queue; a worker that did not come up */

#ifdef __cplusplus
extern "C" {
#endif

typedef struct xwq xwq;

/* One job: a closed GOP (or a whole stream) of one input.  The queue does not look inside; `user` travels with it. */
typedef struct xwq_job {
    int      stream;            /* which input                                                                  */
    int      unit;              /* which GOP of it                                                              */
    uint64_t offset, size;      /* byte range of the unit's NAL units (length prefixes included) in the stream  */
    int      first_picture;     /* number of pictures of the stream before this unit (output position)          */
    int      n_pictures;        /* slice NAL units in the unit                                                  */
    void    *user;
} xwq_job;

xwq *xwq_create(void);
void xwq_destroy(xwq *q);
int  xwq_push(xwq *q, const xwq_job *job);      /* thread-safe; 0 / < 0 after xwq_close                          */
void xwq_close(xwq *q);                         /* no more jobs: poppers drain the queue, then get 0              */
int  xwq_pop(xwq *q, xwq_job *out);             /* blocks; 1: `out` holds a job, 0: closed and empty              */

/* Per-device workers.  init(device, user) -> the worker's state (its xgpu_ctx ...), NULL = the device is unusable (the worker leaves,
   its jobs go to the others); job(state, job) -> 0 or a negative error; fini(state).  Runs until the queue is closed and empty.
   jobs_done[i] = jobs worker i completed.  Returns 0, or the first negative code a job returned (the queue still drains). */
typedef void *(*xwq_init_fn)(int device, void *user);
typedef int   (*xwq_job_fn)(void *state, const xwq_job *job);
typedef void  (*xwq_fini_fn)(void *state);
int  xwq_run(xwq *q, const int *devices, int n_devices, xwq_init_fn init, xwq_job_fn job, xwq_fini_fn fini, void *user, int *jobs_done);

/* Cut a length-prefixed EVC stream (4-byte big-endian NAL size, app/xevd_app.c:52-107) into closed GOPs: a unit starts at every IDR slice NAL
   and takes everything up to the next one.  Parameter sets / APS NAL units before a unit stay where they are - xwq_unit_bytes() prepends
   them.  Returns the number of units (<= max_jobs), -202 for a damaged length prefix, -203 when the stream has more units than max_jobs
   (the array then holds no usable split: call again with a larger one). */
int  xwq_split_gops(const uint8_t *data, size_t size, int stream, xwq_job *jobs, int max_jobs);
/* The bytes a worker decodes for one unit: every SPS / PPS / APS NAL unit of the stream before job->offset (later ones replace earlier
   ones in the parser exactly as they would in a sequential decode), then the unit itself.  Returns the size written, 0 if cap is too small. */
size_t xwq_unit_bytes(const uint8_t *data, size_t size, const xwq_job *job, uint8_t *out, size_t cap);

#ifdef __cplusplus
}
#endif
#endif /* XEVD_WQ_H */
