/*
 * seqoia.h -- drop-in public header of libsqoa_mi355x.so.
 *
 * Source-compatible with the public section of jido/seqoia's single header
 * (reference seqoia.h:288-380): same constants, same sqoa_desc, same four entry
 * points with the same ownership and failure rules.  What differs is where the
 * work happens: the per-pixel opcode loops (reference seqoia.h:530-638 encode,
 * :722-806 decode) run as hand-written gfx950 kernels inside the shared
 * library, so this header carries NO implementation.
 *
 * Callers that did
 *
 *     #define SQOA_IMPLEMENTATION
 *     #include "seqoia.h"
 *
 * (sqoabench.c:31-32, sqoaconv.c:33-34, sqoafuzz.c:15-16) keep compiling
 * unchanged: SQOA_IMPLEMENTATION is accepted and ignored, and the four symbols
 * resolve at link time with  -lsqoa_mi355x .  SQOA_NO_STDIO still hides
 * sqoa_read / sqoa_write (reference seqoia.h:326, :811).
 *
 * Every entry point below cites the reference declaration it replaces.
 * Batch / device-resident entry points live in sqoa_hip.h.
 */
#ifndef SQOA_H
#define SQOA_H

#ifdef __cplusplus
extern "C" {
#endif

/* channel layouts and colour spaces (reference seqoia.h:309-316) */
#define SQOA_CHAN_MONO  1
#define SQOA_CHAN_MONOA 2
#define SQOA_CHAN_RGB   3
#define SQOA_CHAN_RGBA  4
#define SQOA_CHAN_BGR   5 /* encoded without swizzle as 3 (reference seqoia.h:486, :505) */
#define SQOA_CHAN_BGRA  6 /* encoded without swizzle as 4 */
#define SQOA_SRGB   0
#define SQOA_LINEAR 1

/* Image descriptor (reference seqoia.h:318-324).  Field order and types are
 * ABI: callers build it with designated initialisers (sqoabench.c:434-440,
 * sqoaconv.c:84-90).  sizeof == 12. */
typedef struct {
    unsigned int width;
    unsigned int height;
    unsigned char channels;   /* encode: layout of the input pixels; decode: byte from the file header */
    unsigned char colorspace; /* informative only, stored in the header */
    unsigned char qoi_compat; /* encode: 1 = write QOI ("qoif"); decode: set to 1 when the stream is QOI */
} sqoa_desc;

#ifndef SQOA_NO_STDIO
/* Replaces reference seqoia.h:336 (definition :814-836).  Encodes `data` on the
 * GPU and writes the whole stream to `filename`.  Returns bytes written, 0 on
 * any failure. */
int sqoa_write(const char *filename, const void *data, const sqoa_desc *desc);

/* Replaces reference seqoia.h:350 (definition :838-866).  Reads the whole file
 * and decodes it on the GPU.  channels: 0 = as stored, 1..4 = forced output
 * layout.  Returns malloc() memory the caller free()s, NULL on failure. */
void *sqoa_read(const char *filename, sqoa_desc *desc, int channels);
#endif

/* Replaces reference seqoia.h:363 (definition :456-650).  `data` is
 * width*height*{1,2,3,4} tightly packed bytes in host memory (channels 5/6 read
 * as 3/4).  Returns malloc() memory holding the complete stream (header,
 * chunks, 8-byte end marker) and its size in *out_len; the caller free()s it.
 * NULL on invalid arguments, >= 400 Mpixel images, mono + qoi_compat, or when no
 * gfx950 device is usable (there is no CPU fallback). */
void *sqoa_encode(const void *data, const sqoa_desc *desc, int *out_len);

/* Replaces reference seqoia.h:374 (definition :652-809).  Decodes a SQOA or QOI
 * stream held in host memory; *desc receives the header fields.  Returns
 * malloc() memory of width*height*out_channels bytes, NULL on failure.  Unlike
 * the reference, channels < 0 is rejected instead of being undefined. */
void *sqoa_decode(const void *data, int size, sqoa_desc *desc, int channels);

#ifdef __cplusplus
}
#endif
#endif /* SQOA_H */

/* What the reference's implementation section gives its callers besides the four
 * definitions: <stdlib.h> and <string.h> (reference seqoia.h:386-388; sqoafuzz.c:29
 * calls free() on the strength of it) and <stdio.h> (seqoia.h:811-812).  Like the
 * reference's, this part sits outside the include guard. */
#ifdef SQOA_IMPLEMENTATION
#include <stdlib.h>
#include <string.h>
#ifndef SQOA_NO_STDIO
#include <stdio.h>
#endif
#endif
