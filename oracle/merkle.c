/*
 * oracle/merkle.c -- Poseidon2 Merkle commitment over matrix rows, CPU restatement.
 * TEST INFRASTRUCTURE ONLY; PARITY UNPINNED (see oracle/oracle.h).
 *
 * Restates p3-merkle-tree 0.2.1-succinct FieldMerkleTreeMmcs (reference
 * Cargo.lock:4013; reached from crates/guest-prover-sp1/src/sp1.rs:116):
 *   leaf i      = sponge(row i of every matrix of the tallest height, concatenated)
 *   parent      = compress(left, right)
 *   injection   = when a level has as many nodes as a shorter matrix has rows,
 *                 node i = compress(node i, sponge(row i of those matrices))
 *   open/verify = the sibling path, leaf rows re-hashed by the verifier.
 */
#include "oracle.h"
#include <stdlib.h>
#include <string.h>

static void hash_concat_row(const uint32_t* const* mats, const size_t* widths,
                            const int* sel, int nsel, size_t row, uint32_t out[8]) {
    size_t total = 0;
    for (int k = 0; k < nsel; k++) total += widths[sel[k]];
    uint32_t* buf = (uint32_t*)malloc((total ? total : 1) * sizeof(uint32_t));
    size_t pos = 0;
    for (int k = 0; k < nsel; k++) {
        int m = sel[k];
        memcpy(buf + pos, mats[m] + row * widths[m], widths[m] * sizeof(uint32_t));
        pos += widths[m];
    }
    orc_sponge_hash(buf, total, out);
    free(buf);
}

/* rows r0 .. r0 + 7 of the selected matrices, concatenated, hashed eight at a time (poseidon2_x8.c); a single selected matrix is read in place */
static void hash_concat_rows8(const uint32_t* const* mats, const size_t* widths, const int* sel, int nsel, size_t r0, uint32_t* const out[8]) {
    size_t total = 0;
    for (int k = 0; k < nsel; k++) total += widths[sel[k]];
    const uint32_t* in[8];
    uint32_t* buf = NULL;
    if (nsel == 1) {
        for (int l = 0; l < 8; l++) in[l] = mats[sel[0]] + (r0 + (size_t)l) * widths[sel[0]];
    } else {
        buf = (uint32_t*)malloc((total ? total : 1) * 8 * sizeof(uint32_t));
        for (int l = 0; l < 8; l++) {
            size_t pos = 0;
            for (int k = 0; k < nsel; k++) {
                int m = sel[k];
                memcpy(buf + (size_t)l * total + pos, mats[m] + (r0 + (size_t)l) * widths[m], widths[m] * sizeof(uint32_t));
                pos += widths[m];
            }
            in[l] = buf + (size_t)l * total;
        }
    }
    orc_sponge_hash_x8(in, total, out);
    free(buf);
}
static void hash_rows_range(const uint32_t* const* mats, const size_t* widths, const int* sel, int nsel, size_t n, uint32_t* digests) {
    const size_t n8 = orc_simd_enabled() ? n / 8 : 0;
#pragma omp parallel for schedule(static)
    for (size_t b = 0; b < n8; b++) {
        uint32_t* out[8];
        for (int l = 0; l < 8; l++) out[l] = digests + 8 * (8 * b + (size_t)l);
        hash_concat_rows8(mats, widths, sel, nsel, 8 * b, out);
    }
#pragma omp parallel for schedule(static)
    for (size_t r = 8 * n8; r < n; r++) hash_concat_row(mats, widths, sel, nsel, r, digests + 8 * r);
}

void orc_hash_rows(const uint32_t* const* mats, const size_t* widths, int nmats,
                   size_t height, uint32_t* digests) {
    int sel[64];
    for (int i = 0; i < nmats; i++) sel[i] = i;
    hash_rows_range(mats, widths, sel, nmats, height, digests);
}

void orc_merkle_tree_mixed(const uint32_t* const* mats, const size_t* widths,
                           const int* log_heights, int nmats, uint32_t* tree) {
    int log_h = 0;
    for (int i = 0; i < nmats; i++) if (log_heights[i] > log_h) log_h = log_heights[i];
    int sel[64], nsel = 0;
    for (int i = 0; i < nmats; i++) if (log_heights[i] == log_h) sel[nsel++] = i;
    size_t n = (size_t)1 << log_h;
    hash_rows_range(mats, widths, sel, nsel, n, tree);
    uint32_t* prev = tree;
    for (int lvl = log_h - 1; lvl >= 0; lvl--) {
        size_t cnt = (size_t)1 << lvl;
        uint32_t* cur = prev + 16 * cnt;   /* prev has 2*cnt digests */
        nsel = 0;
        for (int i = 0; i < nmats; i++) if (log_heights[i] == lvl) sel[nsel++] = i;
        const size_t c8 = orc_simd_enabled() ? cnt / 8 : 0;
#pragma omp parallel for schedule(static)
        for (size_t b = 0; b < c8; b++) {
            const uint32_t *l8[8], *r8[8];
            uint32_t* o8[8];
            for (int l = 0; l < 8; l++) { const size_t i = 8 * b + (size_t)l; l8[l] = prev + 16 * i; r8[l] = prev + 16 * i + 8; o8[l] = cur + 8 * i; }
            orc_compress_x8(l8, r8, o8);
            if (nsel) {
                uint32_t rh[8][8];
                uint32_t* rp[8];
                const uint32_t* rc[8];
                for (int l = 0; l < 8; l++) { rp[l] = rh[l]; rc[l] = rh[l]; l8[l] = o8[l]; }
                hash_concat_rows8(mats, widths, sel, nsel, 8 * b, rp);
                orc_compress_x8(l8, rc, o8);
            }
        }
#pragma omp parallel for schedule(static)
        for (size_t i = 8 * c8; i < cnt; i++) {
            orc_compress(prev + 16 * i, prev + 16 * i + 8, cur + 8 * i);
            if (nsel) {
                uint32_t rh[8];
                hash_concat_row(mats, widths, sel, nsel, i, rh);
                orc_compress(cur + 8 * i, rh, cur + 8 * i);
            }
        }
        prev = cur;
    }
}

void orc_merkle_tree(const uint32_t* const* mats, const size_t* widths, int nmats,
                     int log_h, uint32_t* tree) {
    int lh[64];
    for (int i = 0; i < nmats; i++) lh[i] = log_h;
    orc_merkle_tree_mixed(mats, widths, lh, nmats, tree);
}

int orc_merkle_verify(const uint32_t root[8], int log_h, size_t index,
                      const uint32_t* const* rows, const size_t* widths, int nmats,
                      const uint32_t* siblings) {
    /* equal-height matrices: `rows[m]` is the opened row of matrix m */
    size_t total = 0;
    for (int m = 0; m < nmats; m++) total += widths[m];
    uint32_t* buf = (uint32_t*)malloc((total ? total : 1) * sizeof(uint32_t));
    size_t pos = 0;
    for (int m = 0; m < nmats; m++) { memcpy(buf + pos, rows[m], widths[m] * 4); pos += widths[m]; }
    uint32_t cur[8];
    orc_sponge_hash(buf, total, cur);
    free(buf);
    for (int lvl = 0; lvl < log_h; lvl++) {
        const uint32_t* sib = siblings + 8 * lvl;
        if ((index >> lvl) & 1) orc_compress(sib, cur, cur);
        else orc_compress(cur, sib, cur);
    }
    return memcmp(cur, root, 32) == 0 ? 0 : 1;
}

/* ---- single-matrix commitment with a selectable hash (hash_width 16: the functions above;
 * 24: Poseidon2 width 24, sponge rate 16, compress(l, r) = permute(l || r || 0^8)[0..8] --
 * RISC Zero's shape) ---- */
void orc_merkle_tree_hw(const uint32_t* mat, size_t width, int log_h, uint32_t* tree, int hash_width) {
    if (hash_width != 24) {
        const uint32_t* mats[1] = {mat}; size_t ws[1] = {width};
        orc_merkle_tree(mats, ws, 1, log_h, tree);
        return;
    }
    size_t rows = (size_t)1 << log_h;
#pragma omp parallel for schedule(static)
    for (size_t r = 0; r < rows; r++) orc_sponge24_hash(mat + r * width, width, 1, tree + 8 * r);
    uint32_t* prev = tree;
    for (int lvl = log_h - 1; lvl >= 0; lvl--) {
        size_t cnt = (size_t)1 << lvl;
        uint32_t* cur = prev + 16 * cnt;
#pragma omp parallel for schedule(static)
        for (size_t i = 0; i < cnt; i++) orc_compress24(prev + 16 * i, prev + 16 * i + 8, cur + 8 * i);
        prev = cur;
    }
}
int orc_merkle_verify_hw(const uint32_t root[8], int log_h, size_t index, const uint32_t* row, size_t width,
                         const uint32_t* siblings, int hash_width) {
    if (hash_width != 24) {
        const uint32_t* rows[1] = {row}; size_t ws[1] = {width};
        return orc_merkle_verify(root, log_h, index, rows, ws, 1, siblings);
    }
    uint32_t cur[8];
    orc_sponge24_hash(row, width, 1, cur);
    for (int lvl = 0; lvl < log_h; lvl++) {
        const uint32_t* sib = siblings + 8 * lvl;
        if ((index >> lvl) & 1) orc_compress24(sib, cur, cur);
        else orc_compress24(cur, sib, cur);
    }
    return memcmp(cur, root, 32) == 0 ? 0 : 1;
}
