/*
 * sqoa_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement of the SQOA/QOI codec of jido/seqoia (reference file
 * /root/reference/seqoia.h, cited below as H:line).  It exists so that the
 * HIP path can be checked byte for byte on any machine, including the GPU box
 * where /root/reference is absent.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * link or load this.  The product library (libsqoa_mi355x.so) never does.
 *
 * Pinning: the restatement is checked against (1) the compiled reference
 * itself (oracle/_ref/libsqoa_ref.so, built by oracle/Makefile straight from
 * /root/reference/seqoia.h) and (2) the golden vectors under tests/golden/
 * that oracle/make_golden.py captured from that compiled reference.
 */
#ifndef SQOA_ORACLE_H
#define SQOA_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Same field order/size as sqoa_desc (H:318-324); kept separate so the oracle
 * never includes the product header. */
typedef struct {
    unsigned int width;
    unsigned int height;
    unsigned char channels;
    unsigned char colorspace;
    unsigned char qoi_compat;
} orc_desc;

/* Worst-case encoded size for a w*h image whose header channel byte is
 * `hdr_channels`: w*h*(ch+1) + 14 + 8 (H:487-489) + 1 for the SQOA start byte
 * the reference forgets to count (H:513). */
size_t orc_encode_bound(unsigned w, unsigned h, unsigned hdr_channels);

/* Encode (H:456-650).  Returns a malloc'd buffer (caller frees) or NULL with
 * the reference's failure rules; *out_len receives the byte count. */
void *orc_encode(const void *pixels, const orc_desc *desc, int *out_len);

/* Decode (H:652-809).  Returns malloc'd pixels or NULL.  Differs from the
 * reference only where the reference has undefined behaviour: channels < 0
 * returns NULL (H:664 only rejects > 4). */
void *orc_decode(const void *data, int size, orc_desc *desc, int channels);

/* Synthetic image classes of SURVEY.md Appendix C. */
enum { ORC_CLASS_MIXED = 0, ORC_CLASS_ICON = 1, ORC_CLASS_PHOTO = 2 };
void orc_synth(uint8_t *dst, unsigned w, unsigned h, unsigned channels,
               int cls, uint32_t seed, uint32_t img);

/* FNV-1a 64 (Appendix C). */
uint64_t orc_fnv1a64(const void *data, size_t n);

#ifdef __cplusplus
}
#endif
#endif
