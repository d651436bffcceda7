/* C caller of include/block_aligner_hip.h, compiled by tests/test_c_abi.py with plain gcc: proves the header is valid C,
 * that the by-value structs and data symbols link, and (on a GPU box) that a C program gets the same answers as the
 * oracle. Prints one line per case: name score query_idx reference_idx cigar. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "block_aligner_hip.h"

static void print_cigar(const Cigar* c) {
    static const char ops[] = " M=XID";
    size_t n = block_len_cigar(c);
    if (n == 0) { printf("-"); return; }
    for (size_t i = 0; i < n; i++) {
        OpLen o = block_get_cigar(c, i);
        printf("%lu%c", (unsigned long)o.len, ops[o.op]);
    }
}

/* Part 1: amino acids, the four handle flavours of the reference header */
static void aa_case(const char* name, const char* q, const char* r, const AAMatrix* m, Gaps g, SizeRange s, int trace, int xdrop, int32_t x) {
    size_t ql = strlen(q), rl = strlen(r);
    PaddedBytes* pq = block_new_padded_aa(ql, s.max);
    PaddedBytes* pr = block_new_padded_aa(rl, s.max);
    block_set_bytes_padded_aa(pq, (const uint8_t*)q, ql, s.max);
    block_set_bytes_padded_aa(pr, (const uint8_t*)r, rl, s.max);
    AlignResult res;
    Cigar* c = block_new_cigar(ql, rl);
    if (trace && xdrop) {
        BlockHandle b = block_new_aa_trace_xdrop(ql, rl, s.max);
        block_align_aa_trace_xdrop(b, pq, pr, m, g, s, x);
        res = block_res_aa_trace_xdrop(b);
        block_cigar_aa_trace_xdrop(b, res.query_idx, res.reference_idx, c);
        block_free_aa_trace_xdrop(b);
    } else if (trace) {
        BlockHandle b = block_new_aa_trace(ql, rl, s.max);
        block_align_aa_trace(b, pq, pr, m, g, s, x);
        res = block_res_aa_trace(b);
        block_cigar_aa_trace(b, res.query_idx, res.reference_idx, c);
        block_free_aa_trace(b);
    } else if (xdrop) {
        BlockHandle b = block_new_aa_xdrop(ql, rl, s.max);
        block_align_aa_xdrop(b, pq, pr, m, g, s, x);
        res = block_res_aa_xdrop(b);
        block_free_aa_xdrop(b);
    } else {
        BlockHandle b = block_new_aa(ql, rl, s.max);
        block_align_aa(b, pq, pr, m, g, s, x);
        res = block_res_aa(b);
        block_free_aa(b);
    }
    printf("%s %d %lu %lu ", name, res.score, (unsigned long)res.query_idx, (unsigned long)res.reference_idx);
    print_cigar(c);
    printf("\n");
    block_free_cigar(c);
    block_free_padded_aa(pq);
    block_free_padded_aa(pr);
}

/* Part 1: sequence to profile */
static void profile_case(const char* name, const char* q, const char* consensus, int8_t match, int8_t mismatch, int8_t gap_open, int8_t gap_extend, SizeRange s) {
    size_t ql = strlen(q), pl = strlen(consensus);
    PaddedBytes* pq = block_new_padded_aa(ql, s.max);
    block_set_bytes_padded_aa(pq, (const uint8_t*)q, ql, s.max);
    AAProfile* p = block_new_aaprofile(pl, s.max, gap_extend);
    for (size_t i = 1; i <= pl; i++)
        for (int ch = 'A'; ch <= 'Z'; ch++) block_set_aaprofile(p, i, (uint8_t)ch, ch == consensus[i - 1] ? match : mismatch);
    block_set_all_gap_open_C_aaprofile(p, gap_open);
    block_set_all_gap_close_C_aaprofile(p, 0);
    block_set_all_gap_open_R_aaprofile(p, gap_open);
    BlockHandle b = block_new_aa_trace(ql, pl, s.max);
    block_align_profile_aa_trace(b, pq, p, s, 0);
    AlignResult res = block_res_aa_trace(b);
    Cigar* c = block_new_cigar(ql, pl);
    block_cigar_aa_trace(b, res.query_idx, res.reference_idx, c);
    printf("%s %d %lu %lu ", name, res.score, (unsigned long)res.query_idx, (unsigned long)res.reference_idx);
    print_cigar(c);
    printf("\n");
    block_free_cigar(c);
    block_free_aa_trace(b);
    block_free_aaprofile(p);
    block_free_padded_aa(pq);
}

/* Part 2: nucleotides through the batch launcher */
static int batch_case(void) {
    static const char* seqs[] = {"TTTTTTTTAAAAAAATTTTTTTTT", "TTAAAAAAATTTTTTTTTTTT",      /* README.md:44-45: q, r */
                                 "ACGTACGTACGTTTACGTACGT", "ACGTACGTACGTACGTACGT",
                                 "", "ACGT"};
    uint8_t pool[256];
    uint64_t off[6]; uint32_t len[6];
    size_t at = 0;
    for (int k = 0; k < 6; k++) { off[k] = at; len[k] = (uint32_t)strlen(seqs[k]); memcpy(pool + at, seqs[k], len[k]); at += len[k]; }
    uint64_t q_off[3] = {off[0], off[2], off[4]}, r_off[3] = {off[1], off[3], off[5]};
    uint32_t q_len[3] = {len[0], len[2], len[4]}, r_len[3] = {len[1], len[3], len[5]};
    Gaps g = {-2, -1};
    SizeRange s = {32, 256};
    AlignResult res[3];
    uint32_t runs[256], cig_len[3];
    if (block_batch_align(BA_KIND_NUC, &NW1, g, s, 0, BA_TRACE | BA_CIGAR_EQ, pool, q_off, q_len, r_off, r_len, 3, res, runs, 256, cig_len)) {
        fprintf(stderr, "batch failed: %s\n", ba_last_error());
        return 1;
    }
    static const char ops[] = " M=XID";
    size_t at_run = 0;
    for (int p = 0; p < 3; p++) {
        printf("batch%d %d %lu %lu ", p, res[p].score, (unsigned long)res[p].query_idx, (unsigned long)res[p].reference_idx);
        if (cig_len[p] == 0) printf("-");
        for (uint32_t k = 0; k < cig_len[p]; k++, at_run++) printf("%u%c", runs[at_run] >> 4, ops[runs[at_run] & 15]);
        printf("\n");
    }
    return 0;
}

int main(int argc, char** argv) {
    if (argc > 1 && strcmp(argv[1], "--link-only") == 0) {   /* no device needed: layout facts a C caller depends on */
        printf("sizeof AlignResult=%zu OpLen=%zu Gaps=%zu SizeRange=%zu\n", sizeof(AlignResult), sizeof(OpLen), sizeof(Gaps), sizeof(SizeRange));
        printf("percent_len %lu %lu\n", (unsigned long)block_percent_len(10000, 0.01f), (unsigned long)block_percent_len(10000, 0.1f));
        printf("statics %p %p %p\n", (const void*)&BLOSUM62, (const void*)&NW1, (const void*)&BYTES1);
        enum Operation e = Eq;   /* the tag name a caller of the reference header uses (c/block_aligner.h:17-57) */
        Operation o = e;
        printf("operation %d %zu\n", (int)o, sizeof(Operation));
        return 0;
    }
    if (ba_device_count() < 1) { fprintf(stderr, "no HIP device\n"); return 2; }
    Gaps g11 = {-11, -1};
    SizeRange s32 = {32, 32}, s16_64 = {16, 64};
    aa_case("aa_global", "AAAAAAAA", "AARAAAA", &BLOSUM62, g11, s32, 0, 0, 0);
    aa_case("aa_trace", "AAAAAAAA", "AARAAAA", &BLOSUM62, g11, s32, 1, 0, 0);
    aa_case("aa_xdrop", "MKVLAARNDCEQGHILKMFPSTWYV", "MKVLAARNDCEQGHILKMFPSTWYVAAAAAAAA", &BLOSUM62, g11, s16_64, 0, 1, 50);
    aa_case("aa_trace_xdrop", "MKVLAARNDCEQGHILKMFPSTWYV", "MKVLARNDCEQGHILKMMFPSTWYV", &BLOSUM50, g11, s16_64, 1, 1, 50);
    profile_case("profile", "ARNDCEQGHIARNDCEQGHI", "ARNDCEQGHIKARNDCEQGHI", 2, -1, -3, -1, s32);
    return batch_case();
}
