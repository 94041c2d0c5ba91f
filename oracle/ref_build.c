/*
 * ref_build.c -- TEST INFRASTRUCTURE ONLY.
 *
 * Translation unit that compiles the UNMODIFIED reference header where it
 * lies (/root/reference/seqoia.h, found through -I, never copied) into
 * oracle/_ref/libsqoa_ref.so with its four entry points renamed ref_sqoa_*,
 * so the real reference can sit beside the product library in one process.
 *
 * SQOA_MALLOC is the reference's own override hook (seqoia.h:390-393); it is
 * used to pad allocations because sqoa_encode writes one byte past max_size
 * in SQOA mode (start byte not counted: seqoia.h:487-489 vs :513).
 */
#include <stdlib.h>
#define SQOA_MALLOC(sz) malloc((size_t)(sz) + 64)
#define SQOA_FREE(p) free(p)
#define sqoa_encode ref_sqoa_encode
#define sqoa_decode ref_sqoa_decode
#define sqoa_write ref_sqoa_write
#define sqoa_read ref_sqoa_read
#define SQOA_IMPLEMENTATION
#include "seqoia.h"
