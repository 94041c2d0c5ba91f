/*
 * sqoa_hip.h -- batch / device-resident entry points of libsqoa_mi355x.so.
 *
 * Additive to seqoia.h: nothing here changes a legacy symbol.  The reference
 * has one image per call on host memory (seqoia.h:363, :374); a GPU needs many
 * independent images per launch and buffers that stay in HBM, so the batch
 * boundary is a *plan*: a table of jobs (one per image) uploaded once, then
 * launched any number of times with no host work besides the launch.
 *
 * Plain C ABI: pointers, sizes, ints.  Device addresses are passed as
 * unsigned long long so the same structs work from cgo / JNI / ctypes.
 * All functions return 0 on success and a negative SQOA_HIP_E* code otherwise;
 * sqoa_hip_last_error() gives the text for the calling thread.
 */
#ifndef SQOA_HIP_H
#define SQOA_HIP_H

#include <stddef.h>
#include "seqoia.h"

#ifdef __cplusplus
extern "C" {
#endif

#define SQOA_HIP_ENOGPU   (-1) /* no usable gfx950 device / HIP runtime failure */
#define SQOA_HIP_EINVAL   (-2)
#define SQOA_HIP_ENOMEM   (-3)
#define SQOA_HIP_ELAUNCH  (-4)

/* One image.  Encode: src = pixels (desc.channels layout, reference
 * seqoia.h:527-542), dst = output slot of at least sqoa_hip_encode_bound()
 * bytes.  Decode: src = stream of src_size bytes (readable up to the next
 * multiple of 16), dst = width*height*out_channels bytes, desc.channels = the
 * `channels` argument of sqoa_decode (0..4), width/height = the shape the caller
 * laid dst out for: a stream whose header says otherwise is SQOA_HIP_ST_BADARG
 * (0 = take whatever the header says; dst_cap still bounds the write). */
typedef struct {
    unsigned long long src;      /* device address */
    unsigned long long dst;      /* device address */
    unsigned int src_size;       /* decode: stream bytes; encode: ignored */
    unsigned int dst_cap;        /* bytes available at dst */
    sqoa_desc desc;
    unsigned int reserved;       /* 0, or SQOA_HIP_JOB_* flags: strip flags for SQOA-native encode (desc.qoi_compat == 0), WAVEONLY for decode */
} sqoa_hip_job;                  /* 40 bytes */

/* Strip jobs: one image encoded by several jobs (several workgroups), each given a stretch of consecutive pixels
 * (desc.width * desc.height = the stretch's pixel count, src = its first pixel).  A stretch must START at a pixel that
 * differs from its predecessor (no run crosses a cut); then the jobs' streams, concatenated in pixel order behind the
 * 15-byte header the caller writes (reference seqoia.h:497-514), are byte-identical to the one-job stream.  For callers
 * that cut images themselves, SQOA-native only; callers that hand over whole images leave `reserved` 0 -- a plan cuts few large
 * images into strips on the device by itself, QOI-compat ones too (it hands the colour index from strip to strip). */
#define SQOA_HIP_JOB_NOHEADER 1u /* no header in front of the ops */
#define SQOA_HIP_JOB_NOEND    2u /* more pixels follow in another job: no end marker, a trailing run is flushed as a run that ends */
#define SQOA_HIP_JOB_PREV     4u /* the first pixel's predecessor is the pixel in front of src (readable), not (0,0,0,255) */
/* Decode jobs: the caller has read the stream's header itself (the legacy sqoa_decode has, reference seqoia.h:672-688) and found
 * an SQOA-native colour stream that goes to 3 or 4 channels.  A plan whose jobs ALL say so launches only the wave-parallel SQOA
 * decoder and not the QOI-compat / mono / exact kernels behind it.  A stream that needs them after all (a header that says
 * otherwise, or an SQOA_OP_REF met on the way: decoder-only op, never emitted by the encoder) comes back with
 * SQOA_HIP_ST_SLOWPATH and nothing decoded; the caller submits it again without the flag. */
#define SQOA_HIP_JOB_WAVEONLY 8u

/* Per-image outcome, written by the kernels. */
typedef struct {
    int len;                     /* encode: stream bytes; decode: pixel bytes; 0 on failure */
    int status;                  /* 0 ok, else SQOA_HIP_ST_* */
    sqoa_desc desc;              /* decode: header fields as sqoa_decode reports them (seqoia.h:673-677) */
} sqoa_hip_result;               /* 20 bytes */

#define SQOA_HIP_ST_OK        0
#define SQOA_HIP_ST_BADARG    1  /* the reference would return NULL (seqoia.h:465-480, :662-688, :705-707) */
#define SQOA_HIP_ST_OVERFLOW  2  /* dst_cap too small */
#define SQOA_HIP_ST_INTERNAL  3  /* chain wait timed out: a bug, never expected */
#define SQOA_HIP_ST_SLOWPATH  4  /* decode: stream needs another kernel (REF op etc.); handled inside the launch -- or, for a plan whose earlier launches
                                  * never needed those kernels, inside sqoa_hip_plan_results (below) --, seen by callers only for SQOA_HIP_JOB_WAVEONLY jobs */

enum { SQOA_HIP_ENCODE = 0, SQOA_HIP_DECODE = 1 };

typedef struct sqoa_hip_plan sqoa_hip_plan;

int sqoa_hip_device_count(void);
const char *sqoa_hip_last_error(void);

/* The allocator of the four legacy entry points (seqoia.h): every buffer sqoa_encode / sqoa_decode / sqoa_read return, and the
 * file buffer inside sqoa_read, comes from `malloc_fn` and is released -- by the CALLER for what is returned -- with `free_fn`.
 * The reference selects these at compile time (#define SQOA_MALLOC / SQOA_FREE, seqoia.h:390-396), which cannot reach a prebuilt
 * library: a caller that built the reference with an arena allocator calls this once, before its first codec call (the two
 * pointers are process-wide and not synchronised against calls in flight).  NULL = back to malloc / free, the default. */
void sqoa_hip_set_allocator(void *(*malloc_fn)(size_t), void (*free_fn)(void *));

/* The default QOI-compat kernels rest on a property of the LDS that the ISA manual does not state -- that the lanes of ONE instruction which
 * meet on an address are applied in ascending lane order (the encoder's index probe, seqoia.h:563-572, is one atomic with return per row;
 * the decoder's index update, seqoia.h:785-787, a plain write).  The library carries order-free forms of those kernels as well and ASKS the
 * device, once per device and process, when the first plan is created on it (a probe of a few hundred microseconds): only a device that
 * passes gets the fast forms.  SQOA_HIP_LDS_ORDER=0 in the environment (read whenever a plan is created or updated) forces the order-free
 * forms, =1 trusts the order without asking.  Output is byte-identical either way on a device that passes.
 * sqoa_hip_lds_order: 1 = launches on `device` take the forms that rest on the order, 0 = the order-free ones, -1 = no such device.
 * sqoa_hip_lds_order_probe: the probe's raw counts on `device` -- out[0] atomic returns out of order, out[1] plain-write winners that are not
 * the highest lane, out[2] lanes checked; returns 0 when it ran. */
int sqoa_hip_lds_order(int device);
int sqoa_hip_lds_order_probe(int device, unsigned int out[3]);

/* Worst-case stream size for one image: w*h*(ch+1)+14+8 (reference
 * seqoia.h:487-489) plus the start byte the reference forgets (seqoia.h:513),
 * rounded up to 256 so consecutive slots stay aligned. 0 if desc is invalid. */
size_t sqoa_hip_encode_bound(const sqoa_desc *desc);

/* Device memory helpers for C callers without another allocator. */
int sqoa_hip_malloc(int device, size_t bytes, unsigned long long *dptr);
int sqoa_hip_free(int device, unsigned long long dptr);
int sqoa_hip_memcpy_h2d(int device, unsigned long long dst, const void *src, size_t bytes);
int sqoa_hip_memcpy_d2h(int device, void *dst, unsigned long long src, size_t bytes);

/* Pinned host memory and stream-ordered copies, for callers that stage host buffers themselves (the legacy entry points
 * of seqoia.h and the directory front-end do): a copy from pinned memory runs at PCIe speed and overlaps the caller's own
 * work; `stream` is a handle from sqoa_hip_stream_create, or a plan's own stream (sqoa_hip_plan_stream). */
int sqoa_hip_host_alloc(size_t bytes, void **p);
int sqoa_hip_host_free(void *p);
int sqoa_hip_stream_create(int device, void **stream);
int sqoa_hip_stream_destroy(int device, void *stream);
int sqoa_hip_stream_sync(int device, void *stream);
int sqoa_hip_memcpy_h2d_async(int device, unsigned long long dst, const void *src, size_t bytes, void *stream);
int sqoa_hip_memcpy_d2h_async(int device, void *dst, unsigned long long src, size_t bytes, void *stream);

/* Plan: uploads the job table, allocates the result table. `kind` is
 * SQOA_HIP_ENCODE or SQOA_HIP_DECODE.
 * A decode plan chooses per stream between one workgroup per stream (a full batch) and three chip-wide passes (few large
 * streams: one workgroup is one compute unit's speed); SQOA_HIP_MULTIPASS=0 / =1 in the environment when the plan is
 * created or updated forces the one or the other for every stream (tests, measurements; the decoded bytes are the same).
 * QOI-compat streams among those few large ones are decoded chip-wide by guess and check; creating or
 * updating such a plan reads one byte of each of these streams where it lies (is any of them QOI-compat framed?) to spare
 * the others three idle launches -- a hint only: streams that arrive later, or other streams of the same sizes written into
 * the same buffers, are decoded correctly whatever it said.
 * REQUIREMENT that follows: the `src` of a decode job that takes the chip-wide route (a job of a plan with few, large streams)
 * must be mapped, readable device memory when sqoa_hip_plan_create / sqoa_hip_plan_update is called (its CONTENT may still be
 * on its way from a producer on another stream: the read is one byte, on the plan's own stream, and not ordered against anybody;
 * what it sees only steers the hint).  Jobs flagged SQOA_HIP_JOB_WAVEONLY are never read at fill time.
 * A decode plan whose 4-channel jobs carry most of its stream bytes runs the wave decoder with twelve stream bytes per lane instead
 * of eight, and one of run-heavy streams (under a stream byte per pixel, where the jobs state width and height) the kernel variant
 * that stores 64 consecutive pixels per instruction; SQOA_HIP_DEC_WIDE / SQOA_HIP_DEC_SWEEP = 0 / 1 in the environment
 * force the one or the other (tests, measurements; the decoded bytes are the same). */
/* The choice a decode plan makes, as a pure host function (no device needed): order[0 .. *n_multipass) = the streams that take
 * the three passes, largest first; order[*n_multipass .. n) = the dispatch order of the streams that get a workgroup each
 * (largest first, every second run of `cus` reversed so that the compute units' loads even out).  `src_sizes[i]` = stream
 * bytes, `cus` = compute units of the device (256), `mode` 0 = cost model, -1 = never multi-pass, 1 = always. */
int sqoa_hip_decode_route(const unsigned int *src_sizes, int n, int cus, int mode, int *order, int *n_multipass);

/* The same for an encode plan's SQOA-native images (`pixels[i]` = width * height): order[0 .. *n_strip_images) = the images the
 * plan cuts into strips on the device (one image is otherwise one workgroup = one compute unit; strips[k] = how many strips
 * order[k] gets), the rest in dispatch order.  SQOA_HIP_STRIPS=0 / =1 in the environment forces never / every image that is
 * large enough; a plan that holds strip jobs of the caller's own (SQOA_HIP_JOB_*) cuts nothing. */
int sqoa_hip_encode_route(const unsigned long long *pixels, int n, int cus, int mode, int *order, int *strips, int *n_strip_images);

int sqoa_hip_plan_create(int device, int kind, const sqoa_hip_job *jobs, int n, sqoa_hip_plan **out);
void sqoa_hip_plan_destroy(sqoa_hip_plan *plan);

/* Replace the job table of an existing plan (n <= the n it was created with):
 * same batch shape, new buffers.  Synchronises the plan's stream first. */
int sqoa_hip_plan_update(sqoa_hip_plan *plan, const sqoa_hip_job *jobs, int n);

/* The plan's own stream (what sqoa_hip_plan_launch(plan, NULL) uses), as a handle for the copy helpers above. */
void *sqoa_hip_plan_stream(sqoa_hip_plan *plan);

/* Enqueue the plan's kernels on `stream` (a hipStream_t, NULL = the plan's own
 * stream).  Asynchronous; no allocation, no host sync: graph-capturable. */
int sqoa_hip_plan_launch(sqoa_hip_plan *plan, void *stream);

/* Diagnostic (tests, tools): the kernel shape the plan's NEXT launch takes for the streams that get a workgroup each -- decode plans: bit 0 many
 * short streams, bit 1 twelve stream bytes per lane, bit 2 position-major stores, bit 3 the balanced kernel, bit 4 its three-channel instance;
 * bit 8: the next launch leaves out the kernels behind the wave decoder (no job of the last fully viewed launch needed them: sqoa_hip_plan_results);
 * bit 9: a part of the list is launched apart, beside the rest, with a shape of its own (a list of mixed output formats, or of mixed content).
 * A decode plan that took the balanced kernel drops it from the launch after sqoa_hip_plan_results has seen, in a full view of a launch's rows,
 * that most of its pixels come from streams that kernel is slow on (two-byte ops throughout: photographs; or more op starts in a chunk than it has
 * slots: ramps, flat images) -- the kernels report what they saw in the rows; where such streams are a twentieth to a half of the pixels they are
 * split off into a launch of their own beside the others; SQOA_HIP_DEC_CONTENT=0 in the environment keeps the first choice. */
int sqoa_hip_plan_shape(const sqoa_hip_plan *plan);

/* Wait for the plan's most recent launch and copy the n results to host memory.  This call is what makes a launch's statuses and
 * lengths FINAL: a decode plan launches two more kernels behind its wave decoder (QOI-compat streams, mono streams, 1- / 2-channel
 * output, SQOA_OP_REF; behind the balanced kernel a third in front of them, for streams with more op starts in a 1024-byte chunk than
 * that kernel has slots) until a launch's rows, read here in full (n = all jobs), show that no job needed them; from then on a launch
 * leaves them out (two launches, ~9 us: a fifth of a plan of 4096 small icons), and if a later launch's rows come back
 * SQOA_HIP_ST_SLOWPATH -- other streams in the same buffers --, this call runs the two kernels before it returns the rows.  A caller
 * that consumes a decode plan's pixels on its own stream without asking for the results (graph replay) sees, for such a stream,
 * what it would see for any failed job: pixels that are not there yet.  SQOA_HIP_DEC_LAZY=0 in the environment: the two kernels
 * with every launch. */
int sqoa_hip_plan_results(sqoa_hip_plan *plan, sqoa_hip_result *out, int n);

/* One-shot forms of the batch extension SURVEY.md section 8(b) asks for ("sqoa_encode_batch / sqoa_decode_batch": an
 * array of descriptors and device buffers in, output slots + lengths out): create a plan on `device`, launch it, wait,
 * copy the n results to `results`, destroy the plan.  Synchronous; callers that repeat a batch shape keep a plan instead. */
int sqoa_hip_encode_batch(int device, const sqoa_hip_job *jobs, int n, sqoa_hip_result *results);
int sqoa_hip_decode_batch(int device, const sqoa_hip_job *jobs, int n, sqoa_hip_result *results);

/* Launch `iters` times back to back on the plan's own stream between two HIP
 * events recorded on that stream; *ms_total receives the elapsed time. */
int sqoa_hip_plan_time(sqoa_hip_plan *plan, int iters, float *ms_total);

/* Exact-size container: after an encode plan has run, copy its n streams out of their
 * worst-case slots into one tightly packed blob; offsets[i] .. offsets[i+1] (n+1 entries, device memory, bytes) is
 * stream i (empty for failed jobs).  *total receives offsets[n].  Asynchronous on `stream` (NULL = the plan's) up to
 * the final read-back of the total, which synchronises. */
int sqoa_hip_pack(sqoa_hip_plan *plan, unsigned long long d_blob, size_t blob_cap, unsigned long long d_offsets,
                  unsigned long long *total, void *stream);

/* Synthetic images of SURVEY.md Appendix C written straight into HBM:
 * n images of w*h*channels bytes, image k = (cls, seed, img0 + k), packed at
 * dst + k*stride. cls: 0 mixed, 1 icon, 2 photo. */
int sqoa_hip_synth(int device, unsigned long long dst, size_t stride, unsigned w, unsigned h,
                   unsigned channels, int cls, unsigned seed, unsigned img0, int n);

/* FNV-1a-64 of n device buffers (one digest per buffer), for size-independent checks. */
int sqoa_hip_fnv1a64(int device, const unsigned long long *dptrs, const unsigned int *lens, int n,
                     unsigned long long *digests);

#ifdef __cplusplus
}
#endif
#endif /* SQOA_HIP_H */
