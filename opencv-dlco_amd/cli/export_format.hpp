// export_format.hpp — the wire format that carries a learned descriptor out of this pipeline:
// the "vgg_generated_XX.i" C header consumed by OpenCV's xfeatures2d VGG descriptor.
//
// Restates the writer of the reference's export-opencv tool (src/export-opencv.cpp:207-391) and
// its pooling-region selection (src/misc.cpp:78-170, SelectPRFilters).  Pure host code: it sits
// after the hot path (the W that pj-learn saves goes in, bytes come out) and has to be
// byte-exact, which tests/test_export_format.py checks against a file the reference ships
// (workspace/opencv/vgg_generated_48.i) and the W it was generated from.
#pragma once

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

namespace dlco_export {

// src/misc.cpp:78-170: keep row i*8+j of PRFilters when w[i] > 0 and the row has a non-zero
// entry, drop exact duplicates (first occurrence wins), sort the rows lexicographically
// (MATLAB's unique(...,'rows') order).  PR is [8*nw][cols]; returns the selected rows.
inline std::vector<float> select_pr_filters(const float *PR, int rows, int cols, const float *w, int nw, int *n_sel)
{
    std::vector<const float *> keep;
    for (int i = 0; i < nw; i++)
        for (int j = 0; j < 8; j++) {
            const int r = i * 8 + j;
            if (r >= rows || !(w[i] > 0.0f)) continue;
            const float *row = PR + (size_t)r * cols;
            bool any = false;
            for (int k = 0; k < cols && !any; k++) any = row[k] != 0.0f;
            if (!any) continue;
            bool inside = false;
            for (const float *o : keep) {
                bool same = true;
                for (int k = 0; k < cols && same; k++) same = row[k] == o[k];
                if (same) { inside = true; break; }
            }
            if (!inside) keep.push_back(row);
        }
    // the reference's insertion sort compares rows element by element; the rows are distinct, so
    // any stable lexicographic sort yields the same order
    std::stable_sort(keep.begin(), keep.end(), [cols](const float *a, const float *b) {
        for (int k = 0; k < cols; k++) {
            if (a[k] == b[k]) continue;
            return a[k] < b[k];
        }
        return false;
    });
    std::vector<float> out((size_t)keep.size() * cols);
    for (size_t i = 0; i < keep.size(); i++) std::memcpy(&out[i * cols], keep[i], (size_t)cols * sizeof(float));
    *n_sel = (int)keep.size();
    return out;
}

// Maximal runs of consecutive non-zero elements of M in flat (row-major) order.
struct Run { long start, count; };
inline std::vector<Run> nonzero_runs(const float *M, size_t n)
{
    std::vector<Run> runs;
    for (size_t e = 0; e < n;) {
        if (M[e] == 0.0f) { e++; continue; }
        size_t b = e;
        while (e < n && M[e] != 0.0f) e++;
        runs.push_back({(long)b, (long)(e - b)});
    }
    return runs;
}

// "indexes & len" array (src/export-opencv.cpp:236-277, and :318-359 for W): the runs as
// (flat start index, run length) pairs — start in lower-case hex, length in upper-case hex, eight
// pairs per line, "};" after the last one.  A matrix without non-zeros produces no bytes at all.
// One behaviour of the reference's writer is part of the format and kept: its scan stops only the
// current row once the last run is out, so for every row BELOW the one holding the last non-zero
// element it emits that last pair and the closing "};" once more (never the case for a learned W,
// whose rows are dense).
inline void write_index_array(FILE *out, const float *M, int rows, int cols)
{
    const std::vector<Run> runs = nonzero_runs(M, (size_t)rows * cols);
    if (runs.empty()) return;
    std::fprintf(out, "{\n ");
    for (size_t i = 0; i < runs.size(); i++) {
        std::fprintf(out, "0x%x,0x%X", (unsigned)runs[i].start, (unsigned)runs[i].count);
        if (i + 1 == runs.size()) break;
        std::fprintf(out, ",");
        if ((i + 1) % 8 == 0) std::fprintf(out, "\n ");
    }
    std::fprintf(out, "\n};\n");
    const Run &last = runs.back();
    const long last_row = (last.start + last.count - 1) / cols;
    for (long r = last_row + 1; r < rows; r++)
        std::fprintf(out, "0x%x,0x%X\n};\n", (unsigned)last.start, (unsigned)last.count);
}

// the non-zero elements themselves as the bit patterns of the floats, eight per line
// (src/export-opencv.cpp:284-306); the opening brace is written even when there is no element
inline void write_value_array(FILE *out, const float *M, int rows, int cols)
{
    std::vector<uint32_t> bits;
    for (size_t e = 0; e < (size_t)rows * cols; e++)
        if (M[e] != 0.0f) { uint32_t b; std::memcpy(&b, &M[e], 4); bits.push_back(b); }
    if (rows * cols > 0) std::fprintf(out, "{\n ");
    for (size_t i = 0; i < bits.size(); i++) {
        std::fprintf(out, "0x%08x", bits[i]);
        if (i + 1 == bits.size()) { std::fprintf(out, "\n};\n"); break; }
        std::fprintf(out, ",");
        if ((i + 1) % 8 == 0) std::fprintf(out, "\n ");
    }
}

// the whole header, src/export-opencv.cpp:207-388
inline void write_vgg_header(FILE *out, const std::string &prg_name, int widx, const std::string &prj_name, const float *sPR,
                             int pr_rows, int pr_cols, const float *W, int w_rows, int w_cols)
{
    std::fprintf(out, "// generated VGG pooling region filters & projection parameters\n");
    std::fprintf(out, "\n");
    std::fprintf(out, "// PR: [%s]#%i\n", prg_name.c_str(), widx);
    std::fprintf(out, "// PJ: [%s]\n", prj_name.c_str());
    std::fprintf(out, "\n");
    std::fprintf(out, "\n");
    std::fprintf(out, "// PR orig rows\n");
    std::fprintf(out, "static const int PRrows = %i;\n", pr_rows);
    std::fprintf(out, "\n");
    std::fprintf(out, "// PR orig cols\n");
    std::fprintf(out, "static const int PRcols = %i;\n", pr_cols);
    std::fprintf(out, "\n");
    std::fprintf(out, "// PR indexes & len\n");
    std::fprintf(out, "static const unsigned int PRidx[] =\n");
    write_index_array(out, sPR, pr_rows, pr_cols);
    std::fprintf(out, "\n");
    std::fprintf(out, "// PR matrix\n");
    std::fprintf(out, "static const unsigned int PR[] =\n");
    write_value_array(out, sPR, pr_rows, pr_cols);
    std::fprintf(out, "\n");
    std::fprintf(out, "\n");
    std::fprintf(out, "// PJ orig rows\n");
    std::fprintf(out, "static const int PJrows = %i;\n", w_rows);
    std::fprintf(out, "\n");
    std::fprintf(out, "// PJ orig cols\n");
    std::fprintf(out, "static const int PJcols = %i;\n", w_cols);
    std::fprintf(out, "\n");
    std::fprintf(out, "// PJ indexes & len\n");
    std::fprintf(out, "static const unsigned int PJidx[] =\n");
    write_index_array(out, W, w_rows, w_cols);
    std::fprintf(out, "\n");
    std::fprintf(out, "// PJ sparse elements\n");
    std::fprintf(out, "static const unsigned int PJ[] =\n");
    write_value_array(out, W, w_rows, w_cols);
}

}  // namespace dlco_export
