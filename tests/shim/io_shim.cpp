// Test-only shim around the command-line tools' I/O layer (opencv-dlco_amd/cli/dlco_io.hpp, header-only,
// no HIP): lets the test-suite (a) read datasets of an HDF5 file through the product's reader and
// (b) write an input file the way the reference's producer does — "Label" u8 [N,1] and "Distance"
// f32 [N,F], both chunked {128,1} with gzip level 9 (src/comp-uprjdists.cpp:254,289-290) — since
// h5py is not available.  Built by the tests with g++.
#include "../../opencv-dlco_amd/cli/dlco_io.hpp"

#include <cstring>

extern "C" {

// returns the number of dimensions (<= 4) or a negative code; `out` receives up to `cap` floats
int shim_read_f32(const char *path, const char *name, float *out, size_t cap, size_t *shape)
{
    try {
        std::vector<size_t> sh;
        std::vector<float> v;
        dlco_io::read_dataset<float>(path, name, sh, v);
        if (sh.size() > 4 || v.size() > cap) return -2;
        for (size_t i = 0; i < sh.size(); i++) shape[i] = sh[i];
        std::memcpy(out, v.data(), v.size() * sizeof(float));
        return (int)sh.size();
    } catch (const std::exception &e) {
        std::fprintf(stderr, "shim_read_f32: %s\n", e.what());
        return -1;
    }
}

int shim_hdf5_available() { return dlco_io::h5().load() ? 1 : 0; }

// the producer's layout: chunk {chunk_rows, chunk_cols}, deflate `gzip` (0 = none)
int shim_write_unproj(const char *path, const float *D, const unsigned char *L, size_t N, size_t F, size_t chunk_rows,
                      size_t chunk_cols, int gzip)
{
    typedef dlco_io::H5::hid_t hid_t;
    dlco_io::H5 &h = dlco_io::h5();
    if (!h.load()) return -1;
    auto sym = [&](const char *n) { return dlsym(h.h, n); };
    auto H5Pcreate = reinterpret_cast<hid_t (*)(hid_t)>(sym("H5Pcreate"));
    auto H5Pset_chunk = reinterpret_cast<int (*)(hid_t, int, const unsigned long long *)>(sym("H5Pset_chunk"));
    auto H5Pset_deflate = reinterpret_cast<int (*)(hid_t, unsigned)>(sym("H5Pset_deflate"));
    auto H5Pclose = reinterpret_cast<int (*)(hid_t)>(sym("H5Pclose"));
    hid_t *dcpl_cls = reinterpret_cast<hid_t *>(sym("H5P_CLS_DATASET_CREATE_ID_g"));
    if (!H5Pcreate || !H5Pset_chunk || !H5Pset_deflate || !H5Pclose || !dcpl_cls) return -2;
    const hid_t f = h.H5Fcreate(path, 2 /* H5F_ACC_TRUNC */, 0, 0);
    if (f < 0) return -3;
    int rc = 0;
    for (int which = 0; which < 2 && rc == 0; which++) {
        const unsigned long long dims[2] = {N, which == 0 ? 1ULL : F};
        unsigned long long chunk[2] = {chunk_rows < N ? chunk_rows : N, chunk_cols < dims[1] ? chunk_cols : dims[1]};
        const hid_t p = H5Pcreate(*dcpl_cls);
        H5Pset_chunk(p, 2, chunk);
        if (gzip > 0) H5Pset_deflate(p, (unsigned)gzip);
        const hid_t s = h.H5Screate_simple(2, dims, nullptr);
        const hid_t type = which == 0 ? h.native_uchar : h.native_float;
        const hid_t d = h.H5Dcreate2(f, which == 0 ? "Label" : "Distance", type, s, 0, p, 0);
        if (d < 0) rc = -4;
        else if (h.H5Dwrite(d, type, 0, 0, 0, which == 0 ? (const void *)L : (const void *)D) < 0) rc = -5;
        if (d >= 0) h.H5Dclose(d);
        h.H5Sclose(s);
        H5Pclose(p);
    }
    h.H5Fclose(f);
    return rc;
}


// ---- the image-set / filter-bank side of the pipeline (comp-uprjdists, comp-fulldists inputs) -------------------
// writes one dataset of any rank with the given native type id: 0 = f32, 1 = u8, 2 = i32 (contiguous layout)
int shim_write_nd(const char *path, int create, const char *name, int type, const void *data, int nd, const size_t *shape)
{
    typedef dlco_io::H5::hid_t hid_t;
    dlco_io::H5 &h = dlco_io::h5();
    if (!h.load()) return -1;
    const hid_t f = create ? h.H5Fcreate(path, 2 /* H5F_ACC_TRUNC */, 0, 0) : h.H5Fopen(path, 1 /* H5F_ACC_RDWR */, 0);
    if (f < 0) return -3;
    unsigned long long dims[4];
    for (int i = 0; i < nd; i++) dims[i] = shape[i];
    const hid_t s = h.H5Screate_simple(nd, dims, nullptr);
    const hid_t t = type == 0 ? h.native_float : (type == 1 ? h.native_uchar : h.native_int);
    const hid_t d = h.H5Dcreate2(f, name, t, s, 0, 0, 0);
    int rc = 0;
    if (d < 0) rc = -4;
    else if (h.H5Dwrite(d, t, 0, 0, 0, data) < 0) rc = -5;
    if (d >= 0) h.H5Dclose(d);
    h.H5Sclose(s);
    h.H5Fclose(f);
    return rc;
}

// the product's block-of-rows writer (dlco_io::RowStream): D [N,F] + L [N] written `block` rows at a time into chunked
// {chunk_rows, 1} deflate-`gzip` datasets (what comp-uprjdists / comp-fulldists do); `stop_after` > 0 ends the run early
// to show that an interrupted run keeps the rows written so far
int shim_stream_unproj(const char *path, const float *D, const unsigned char *L, size_t N, size_t F, size_t block, size_t chunk_rows, int gzip,
                       size_t stop_after)
{
    try {
        dlco_io::Writer w(path);
        dlco_io::RowStream<uint8_t> ls(w, "Label", N, 1, chunk_rows, 1, (unsigned)gzip);
        dlco_io::RowStream<float> ds(w, "Distance", N, F, chunk_rows, 1, (unsigned)gzip);
        for (size_t r0 = 0; r0 < N; r0 += block) {
            if (stop_after && r0 >= stop_after) break;
            const size_t n = block < N - r0 ? block : N - r0;
            ls.write_rows(r0, n, L + r0);
            ds.write_rows(r0, n, D + r0 * F);
        }
        return 0;
    } catch (const std::exception &e) {
        std::fprintf(stderr, "shim_stream_unproj: %s\n", e.what());
        return -1;
    }
}

int shim_read_i32(const char *path, const char *name, int32_t *out, size_t cap, size_t *shape)
{
    try {
        std::vector<size_t> sh;
        std::vector<int32_t> v;
        dlco_io::read_dataset<int32_t>(path, name, sh, v);
        if (sh.size() > 4 || v.size() > cap) return -2;
        for (size_t i = 0; i < sh.size(); i++) shape[i] = sh[i];
        std::memcpy(out, v.data(), v.size() * sizeof(int32_t));
        return (int)sh.size();
    } catch (const std::exception &e) {
        std::fprintf(stderr, "shim_read_i32: %s\n", e.what());
        return -1;
    }
}

int shim_read_u8(const char *path, const char *name, unsigned char *out, size_t cap, size_t *shape)
{
    try {
        std::vector<size_t> sh;
        std::vector<uint8_t> v;
        dlco_io::read_dataset<uint8_t>(path, name, sh, v);
        if (sh.size() > 4 || v.size() > cap) return -2;
        for (size_t i = 0; i < sh.size(); i++) shape[i] = sh[i];
        std::memcpy(out, v.data(), v.size());
        return (int)sh.size();
    } catch (const std::exception &e) {
        std::fprintf(stderr, "shim_read_u8: %s\n", e.what());
        return -1;
    }
}

}
