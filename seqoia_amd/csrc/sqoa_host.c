/*
 * sqoa_host.c -- the four legacy entry points of seqoia.h, in plain C.
 *
 * This is the drop-in boundary: same signatures, same malloc()/free()
 * ownership, same NULL / 0 failure rules as the reference (seqoia.h:336, :350,
 * :363, :374; definitions :456-650, :652-809, :814-866).  Everything per-pixel
 * happens on the GPU through the batch layer (sqoa_hip.h) with a batch of one:
 * copy in, one launch, copy the exact-size result out.  That round trip is
 * PCIe-bound and exists for drop-in correctness; throughput work uses plans on
 * device-resident buffers.
 *
 * There is NO CPU fallback: without a usable gfx950 device every entry point
 * fails (NULL / 0), silently like the reference; SQOA_HIP_VERBOSE=1 makes the
 * first failure say why on stderr.
 */
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/seqoia.h"
#include "../../include/sqoa_hip.h"

#define SQOA_PIXELS_MAX 400000000u /* reference seqoia.h:432 */

/* The reference takes its allocator at COMPILE time (#define SQOA_MALLOC / SQOA_FREE in front of the implementation,
 * seqoia.h:390-396): every buffer it returns, and the file buffer of sqoa_read, comes from it.  A prebuilt library cannot see
 * the caller's macros; this is the run-time counterpart (sqoa_hip.h).  Default: malloc / free, as the reference's default. */
static void *(*g_malloc)(size_t) = malloc;
static void (*g_free)(void *) = free;

void sqoa_hip_set_allocator(void *(*malloc_fn)(size_t), void (*free_fn)(void *)) {
    g_malloc = malloc_fn ? malloc_fn : malloc;
    g_free = free_fn ? free_fn : free;
}

/* Per-thread device context: the reference is re-entrant (only a const table
 * at seqoia.h:439), so concurrent callers must not share mutable state. */
typedef struct {
    int device;
    sqoa_hip_plan *plan[2];
    sqoa_hip_job last[2]; /* the job each plan holds: a caller that repeats a call (sqoabench does) skips the upload */
    int have_last[2];
    unsigned long long d_in, d_out;
    size_t cap_in, cap_out;
} host_ctx;

static pthread_key_t g_key;
static pthread_once_t g_once = PTHREAD_ONCE_INIT;

static void ctx_free(void *p) {
    host_ctx *c = p;
    if (!c) return;
    for (int k = 0; k < 2; k++)
        if (c->plan[k]) sqoa_hip_plan_destroy(c->plan[k]);
    if (c->d_in) sqoa_hip_free(c->device, c->d_in);
    if (c->d_out) sqoa_hip_free(c->device, c->d_out);
    free(c);
}

static void key_init(void) { pthread_key_create(&g_key, ctx_free); }

/* The reference library never prints (failure is NULL / 0, seqoia.h:333-334, :343-344, :357-358, :368-369) and neither
 * does this one unless SQOA_HIP_VERBOSE is set in the environment: then the first failure says why, once. */
static void complain(const char *what) {
    static int said = 0;
    const char *v = getenv("SQOA_HIP_VERBOSE");
    if (v && *v && *v != '0' && !__sync_lock_test_and_set(&said, 1))
        fprintf(stderr, "libsqoa_mi355x: %s: %s (no CPU fallback)\n", what, sqoa_hip_last_error());
}

static host_ctx *ctx_get(void) {
    pthread_once(&g_once, key_init);
    host_ctx *c = pthread_getspecific(g_key);
    if (c) return c;
    if (sqoa_hip_device_count() <= 0) {
        complain("no HIP device");
        return NULL;
    }
    c = calloc(1, sizeof *c);
    if (!c) return NULL;
    const char *dev = getenv("SQOA_HIP_DEVICE");
    c->device = dev ? atoi(dev) : 0;
    pthread_setspecific(g_key, c);
    return c;
}

static int grow(host_ctx *c, unsigned long long *d, size_t *cap, size_t need) {
    if (*cap >= need) return 0;
    if (*d) sqoa_hip_free(c->device, *d);
    *d = 0;
    *cap = 0;
    need = (need + (1u << 20)) & ~(size_t)((1u << 20) - 1);
    if (sqoa_hip_malloc(c->device, need, d)) return -1;
    *cap = need;
    return 0;
}

/* one job through the thread's plan of the given kind */
static int run_job(host_ctx *c, int kind, const sqoa_hip_job *job, sqoa_hip_result *res) {
    if (!c->plan[kind]) {
        if (sqoa_hip_plan_create(c->device, kind, job, 1, &c->plan[kind])) { complain("plan"); return -1; }
    } else if (!c->have_last[kind] || memcmp(&c->last[kind], job, sizeof *job)) {
        c->have_last[kind] = 0;
        if (sqoa_hip_plan_update(c->plan[kind], job, 1)) { complain("plan update"); return -1; }
    }
    c->last[kind] = *job;
    c->have_last[kind] = 1;
    if (sqoa_hip_plan_launch(c->plan[kind], NULL)) { complain("launch"); return -1; }
    if (sqoa_hip_plan_results(c->plan[kind], res, 1)) { complain("results"); return -1; }
    return 0;
}

void *sqoa_encode(const void *data, const sqoa_desc *desc, int *out_len) {
    /* reference seqoia.h:465-480 */
    if (!data || !out_len || !desc || desc->width == 0 || desc->height == 0 || desc->channels < 1 ||
        desc->channels > 6 || desc->colorspace > 1 || desc->height >= SQOA_PIXELS_MAX / desc->width)
        return NULL;
    if (desc->channels < 3 && desc->qoi_compat) return NULL;
    host_ctx *c = ctx_get();
    if (!c) return NULL;

    const unsigned stride = (desc->channels < 3 ? 1u : 3u) + ((desc->channels & 1) ? 0u : 1u);
    const size_t in_bytes = (size_t)desc->width * desc->height * stride;
    const size_t bound = sqoa_hip_encode_bound(desc);
    if (grow(c, &c->d_in, &c->cap_in, in_bytes) || grow(c, &c->d_out, &c->cap_out, bound)) return NULL;
    if (sqoa_hip_memcpy_h2d(c->device, c->d_in, data, in_bytes)) return NULL;
    /* one job: a plan that holds a lone large SQOA-native image cuts it into strips on the device (sqoa_strips.hip), i.e.
     * the whole chip encodes it; the complete stream, header included, ends up at dst */
    sqoa_hip_job job;
    memset(&job, 0, sizeof job);
    job.src = c->d_in;
    job.dst = c->d_out;
    job.dst_cap = bound > 0xffffffffu ? 0xffffffffu : (unsigned)bound;
    job.desc = *desc;
    sqoa_hip_result res;
    if (run_job(c, SQOA_HIP_ENCODE, &job, &res) || res.status != SQOA_HIP_ST_OK || res.len <= 0) return NULL;
    unsigned char *out = g_malloc((size_t)res.len); /* SQOA_MALLOC, seqoia.h:492; the caller frees: sqoabench.c:537, seqoia.h:834 */
    if (!out) return NULL;
    if (sqoa_hip_memcpy_d2h(c->device, out, c->d_out, (size_t)res.len)) { g_free(out); return NULL; }
    *out_len = res.len;
    return out;
}

static unsigned be32(const unsigned char *b) {
    return (unsigned)b[0] << 24 | (unsigned)b[1] << 16 | (unsigned)b[2] << 8 | b[3];
}

void *sqoa_decode(const void *data, int size, sqoa_desc *desc, int channels) {
    /* reference seqoia.h:662-668; channels < 0 is undefined there, NULL here */
    if (!data || !desc || channels > 4 || channels < 0 || size < 14 + 8) return NULL;
    const unsigned char *b = data;
    const unsigned magic = be32(b);
    /* the reference fills *desc before it validates (seqoia.h:672-677) */
    desc->width = be32(b + 4);
    desc->height = be32(b + 8);
    desc->channels = b[12];
    desc->colorspace = b[13];
    desc->qoi_compat = b[14] != 0x31;
    /* seqoia.h:679-688 */
    if (desc->width == 0 || desc->height == 0 || desc->channels < 1 || desc->channels > 6 || desc->colorspace > 1 ||
        !(magic == 0x716f6966u || magic == 0x53716f61u) || (magic == 0x716f6966u && !desc->qoi_compat) ||
        desc->height >= SQOA_PIXELS_MAX / desc->width)
        return NULL;
    const int col = desc->channels < 3 ? 1 : 3;
    const int out_ch = channels ? channels : col + ((desc->channels & 1) ? 0 : 1); /* seqoia.h:699-702 */

    host_ctx *c = ctx_get();
    if (!c) return NULL;
    const size_t out_bytes = (size_t)desc->width * desc->height * out_ch;
    if (grow(c, &c->d_in, &c->cap_in, (size_t)size + 16) || grow(c, &c->d_out, &c->cap_out, out_bytes)) return NULL;
    if (sqoa_hip_memcpy_h2d(c->device, c->d_in, data, (size_t)size)) return NULL;

    sqoa_hip_job job;
    memset(&job, 0, sizeof job);
    job.src = c->d_in;
    job.dst = c->d_out;
    job.src_size = (unsigned)size;
    job.dst_cap = out_bytes > 0xffffffffu ? 0xffffffffu : (unsigned)out_bytes;
    job.desc.width = desc->width;
    job.desc.height = desc->height;
    job.desc.channels = (unsigned char)channels;
    /* The header has been read right here: an SQOA-native colour stream that goes to 3 or 4 channels needs only the wave-parallel
     * decoder, and the plan is told so (it then launches nothing else).  A crafted stream with an SQOA_OP_REF in it (seqoia.h:729-738)
     * comes back undecoded with SLOWPATH and takes the call again through every kernel. */
    if (!desc->qoi_compat && desc->channels >= 3 && out_ch >= 3) job.reserved = SQOA_HIP_JOB_WAVEONLY;
    sqoa_hip_result res;
    if (run_job(c, SQOA_HIP_DECODE, &job, &res)) return NULL;
    if (job.reserved && res.status == SQOA_HIP_ST_SLOWPATH) {
        job.reserved = 0;
        if (run_job(c, SQOA_HIP_DECODE, &job, &res)) return NULL;
    }
    if (res.status != SQOA_HIP_ST_OK || (size_t)res.len != out_bytes) return NULL;

    void *out = g_malloc(out_bytes); /* SQOA_MALLOC, seqoia.h:710; the caller frees: sqoabench.c:454, :493 */
    if (!out) return NULL;
    if (sqoa_hip_memcpy_d2h(c->device, out, c->d_out, out_bytes)) { g_free(out); return NULL; }
    return out;
}

#ifndef SQOA_NO_STDIO

int sqoa_write(const char *filename, const void *data, const sqoa_desc *desc) { /* seqoia.h:814-836 */
    FILE *f = fopen(filename, "wb");
    if (!f) return 0;
    int size = 0;
    void *enc = sqoa_encode(data, desc, &size);
    if (!enc) { fclose(f); return 0; }
    fwrite(enc, 1, (size_t)size, f);
    fflush(f);
    const int err = ferror(f);
    fclose(f);
    g_free(enc); /* SQOA_FREE, seqoia.h:834 */
    return err ? 0 : size;
}

void *sqoa_read(const char *filename, sqoa_desc *desc, int channels) { /* seqoia.h:838-866 */
    FILE *f = fopen(filename, "rb");
    if (!f) return NULL;
    fseek(f, 0, SEEK_END);
    const long size = ftell(f);
    if (size <= 0 || size > 0x7fffffffL || fseek(f, 0, SEEK_SET) != 0) { fclose(f); return NULL; }
    void *data = g_malloc((size_t)size); /* SQOA_MALLOC, seqoia.h:849 */
    if (!data) { fclose(f); return NULL; }
    const size_t got = fread(data, 1, (size_t)size, f);
    fclose(f);
    void *px = got == (size_t)size ? sqoa_decode(data, (int)size, desc, channels) : NULL;
    g_free(data); /* SQOA_FREE, seqoia.h:864 */
    return px;
}

#endif /* SQOA_NO_STDIO */
