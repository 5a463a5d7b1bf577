/*
 * tgsf_oracle.h -- TEST INFRASTRUCTURE ONLY (see tgsf_oracle.c).
 * CPU restatement of TGSFilter's per-read hot path used as the parity checker.
 */
#ifndef TGSF_ORACLE_H
#define TGSF_ORACLE_H
#include <stdint.h>
#include "../include/tgsf.h"
#ifdef __cplusplus
extern "C" {
#endif

/* What TGSFilter reads out of an EdlibAlignResult (include/edlib.h:162-218). */
typedef struct orc_alignment {
    int  edit_distance;      /* -1: nothing within k */
    int  num_locations;
    int* starts;             /* malloc'ed, num_locations entries */
    int* ends;
    int  alignment_length;
} orc_alignment;

/* edlibAlign(q, Q, t, T, edlibNewAlignConfig(k, EDLIB_MODE_HW, EDLIB_TASK_PATH, NULL, 0)) */
int  orc_align_hw(const uint8_t* q, int Q, const uint8_t* t, int T, int k, orc_alignment* out);
void orc_alignment_free(orc_alignment* a);

/* filter_sequence (src/TGSFilter.cpp:1939-2061) over a CSR batch held in host memory.
 * ctr: tgsf_ctr_len(bc_len, n_bins) words, accumulated into (not zeroed). */
int orc_filter_batch(const tgsf_params* p, const tgsf_batch_in* in, tgsf_batch_out* out,
                     uint64_t* ctr, uint32_t n_bins);

#ifdef __cplusplus
}
#endif
#endif
