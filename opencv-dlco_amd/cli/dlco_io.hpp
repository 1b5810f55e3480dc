// dlco_io.hpp — file I/O of the command-line tools.
//
// The reference exchanges data through HDF5 files with fixed dataset names
// (src/pj-learn.cpp:173-212 reads "Distance" f32 [N,F] and "Label" u8 [N,1];
// :592-597 writes "W" and "A"; src/comp-uprjdists.cpp:254-290 produces the input).
// HDF5 is reached through dlopen of libhdf5 (no build-time dependency); a path that does not
// end in .h5/.hdf5 is treated as a directory of .npy files with the same dataset names
// (Distance.npy, Label.npy, W.npy, A.npy), so the tools also run where HDF5 is absent.
#pragma once

#include <dlfcn.h>
#include <sys/stat.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

namespace dlco_io {

inline bool is_h5(const std::string &p)
{
    auto ends = [&](const char *s) { const size_t n = std::strlen(s); return p.size() >= n && p.compare(p.size() - n, n, s) == 0; };
    struct stat st;
    if (::stat(p.c_str(), &st) == 0 && S_ISDIR(st.st_mode)) return false;      // a directory of .npy files, whatever its name
    return ends(".h5") || ends(".hdf5") || ends(".hdf");
}

// ---------------------------------------------------------------------------------- HDF5 (dlopen)
struct H5 {
    typedef int64_t hid_t;
    void *h = nullptr;
    int (*H5open)() = nullptr;
    hid_t (*H5Fopen)(const char *, unsigned, hid_t) = nullptr;
    hid_t (*H5Fcreate)(const char *, unsigned, hid_t, hid_t) = nullptr;
    int (*H5Fclose)(hid_t) = nullptr;
    hid_t (*H5Dopen2)(hid_t, const char *, hid_t) = nullptr;
    hid_t (*H5Dcreate2)(hid_t, const char *, hid_t, hid_t, hid_t, hid_t, hid_t) = nullptr;
    hid_t (*H5Dget_space)(hid_t) = nullptr;
    int (*H5Sget_simple_extent_ndims)(hid_t) = nullptr;
    int (*H5Sget_simple_extent_dims)(hid_t, unsigned long long *, unsigned long long *) = nullptr;
    hid_t (*H5Screate_simple)(int, const unsigned long long *, const unsigned long long *) = nullptr;
    int (*H5Dread)(hid_t, hid_t, hid_t, hid_t, hid_t, void *) = nullptr;
    int (*H5Dwrite)(hid_t, hid_t, hid_t, hid_t, hid_t, const void *) = nullptr;
    int (*H5Dclose)(hid_t) = nullptr;
    int (*H5Sclose)(hid_t) = nullptr;
    int (*H5Lexists)(hid_t, const char *, hid_t) = nullptr;
    int (*H5Eset_auto2)(hid_t, void *, void *) = nullptr;
    hid_t native_float = -1, native_uchar = -1, native_int = -1;

    bool load()
    {
        if (h) return true;
        const char *cands[] = {getenv("DLCO_HDF5_LIB"), "libhdf5.so", "libhdf5_serial.so", "/opt/conda/lib/libhdf5.so",
                               "/usr/lib/x86_64-linux-gnu/libhdf5_serial.so", "/usr/lib/x86_64-linux-gnu/hdf5/serial/libhdf5.so"};
        for (const char *c : cands) {
            if (!c || !*c) continue;
            h = dlopen(c, RTLD_NOW | RTLD_GLOBAL);
            if (h) break;
        }
        if (!h) return false;
#define DLCO_H5SYM(n) n = reinterpret_cast<decltype(n)>(dlsym(h, #n)); if (!n) return false;
        DLCO_H5SYM(H5open) DLCO_H5SYM(H5Fopen) DLCO_H5SYM(H5Fcreate) DLCO_H5SYM(H5Fclose) DLCO_H5SYM(H5Dopen2)
        DLCO_H5SYM(H5Dcreate2) DLCO_H5SYM(H5Dget_space) DLCO_H5SYM(H5Sget_simple_extent_ndims)
        DLCO_H5SYM(H5Sget_simple_extent_dims) DLCO_H5SYM(H5Screate_simple) DLCO_H5SYM(H5Dread) DLCO_H5SYM(H5Dwrite)
        DLCO_H5SYM(H5Dclose) DLCO_H5SYM(H5Sclose) DLCO_H5SYM(H5Lexists) DLCO_H5SYM(H5Eset_auto2)
#undef DLCO_H5SYM
        // hid_t is a 64-bit integer since HDF5 1.10; an older library would be called with the wrong ABI
        auto getver = reinterpret_cast<int (*)(unsigned *, unsigned *, unsigned *)>(dlsym(h, "H5get_libversion"));
        unsigned maj = 0, min = 0, rel = 0;
        if (!getver || getver(&maj, &min, &rel) < 0 || maj < 1 || (maj == 1 && min < 10)) {
            std::fprintf(stderr, "dlco_io: libhdf5 %u.%u.%u is older than 1.10 (32-bit hid_t): not used\n", maj, min, rel);
            h = nullptr;
            return false;
        }
        H5open();
        H5Eset_auto2(0, nullptr, nullptr);
        hid_t *pf = reinterpret_cast<hid_t *>(dlsym(h, "H5T_NATIVE_FLOAT_g"));
        hid_t *pu = reinterpret_cast<hid_t *>(dlsym(h, "H5T_NATIVE_UCHAR_g"));
        hid_t *pi = reinterpret_cast<hid_t *>(dlsym(h, "H5T_NATIVE_INT_g"));
        if (!pf || !pu || !pi) return false;
        native_float = *pf; native_uchar = *pu; native_int = *pi;
        return true;
    }
};

inline H5 &h5()
{
    static H5 inst;
    return inst;
}

// ---------------------------------------------------------------------------------- .npy (v1.0)
inline void npy_write(const std::string &path, const void *data, const char *descr, size_t elem, const std::vector<size_t> &shape)
{
    FILE *f = std::fopen(path.c_str(), "wb");
    if (!f) throw std::runtime_error("cannot write " + path);
    std::string dict = std::string("{'descr': '") + descr + "', 'fortran_order': False, 'shape': (";
    for (size_t i = 0; i < shape.size(); i++) dict += std::to_string(shape[i]) + (shape.size() == 1 || i + 1 < shape.size() ? "," : "");
    dict += "), }";
    while ((10 + dict.size() + 1) % 64 != 0) dict += ' ';
    dict += '\n';
    const unsigned char magic[8] = {0x93, 'N', 'U', 'M', 'P', 'Y', 1, 0};
    const uint16_t hl = (uint16_t)dict.size();
    std::fwrite(magic, 1, 8, f);
    std::fwrite(&hl, 2, 1, f);
    std::fwrite(dict.data(), 1, dict.size(), f);
    size_t n = elem;
    for (size_t s : shape) n *= s;
    if (n && std::fwrite(data, 1, n, f) != n) { std::fclose(f); throw std::runtime_error("short write " + path); }
    std::fclose(f);
}

inline void npy_read(const std::string &path, const char *want_descr, size_t elem, std::vector<size_t> &shape, std::vector<char> &bytes)
{
    FILE *f = std::fopen(path.c_str(), "rb");
    if (!f) throw std::runtime_error("cannot open " + path);
    unsigned char magic[10];
    if (std::fread(magic, 1, 10, f) != 10 || std::memcmp(magic, "\x93NUMPY", 6) != 0) { std::fclose(f); throw std::runtime_error(path + ": not a .npy file"); }
    size_t hl = magic[8] | (magic[9] << 8);
    if (magic[6] >= 2) {                       // v2/v3: 4-byte header length
        unsigned char more[2];
        if (std::fread(more, 1, 2, f) != 2) { std::fclose(f); throw std::runtime_error(path + ": truncated"); }
        hl |= ((size_t)more[0] << 16) | ((size_t)more[1] << 24);
    }
    std::string dict(hl, ' ');
    if (std::fread(&dict[0], 1, hl, f) != hl) { std::fclose(f); throw std::runtime_error(path + ": truncated header"); }
    if (dict.find(std::string("'") + want_descr + "'") == std::string::npos && dict.find(std::string("'|") + (want_descr + 1) + "'") == std::string::npos)
        { std::fclose(f); throw std::runtime_error(path + ": expected dtype " + want_descr + ", header is " + dict); }
    if (dict.find("'fortran_order': False") == std::string::npos) { std::fclose(f); throw std::runtime_error(path + ": fortran order not supported"); }
    const size_t a = dict.find("'shape': ("), b = dict.find(')', a);
    shape.clear();
    size_t pos = a + 10;
    while (pos < b) {
        while (pos < b && (dict[pos] == ' ' || dict[pos] == ',')) pos++;
        if (pos >= b) break;
        shape.push_back(std::strtoull(dict.c_str() + pos, nullptr, 10));
        while (pos < b && dict[pos] != ',') pos++;
    }
    size_t n = elem;
    for (size_t s : shape) n *= s;
    bytes.resize(n);
    if (n && std::fread(bytes.data(), 1, n, f) != n) { std::fclose(f); throw std::runtime_error(path + ": truncated data"); }
    std::fclose(f);
}

// ---------------------------------------------------------------------------------- datasets
template <typename T> struct ElemType;
template <> struct ElemType<float>   { static const char *npy() { return "<f4"; } static H5::hid_t h5t(H5 &L) { return L.native_float; } };
template <> struct ElemType<uint8_t> { static const char *npy() { return "|u1"; } static H5::hid_t h5t(H5 &L) { return L.native_uchar; } };
template <> struct ElemType<int32_t> { static const char *npy() { return "<i4"; } static H5::hid_t h5t(H5 &L) { return L.native_int; } };

// Reads dataset `name` as f32, u8 or i32 into `out`; shape receives its dimensions.
template <typename T>
void read_dataset(const std::string &path, const char *name, std::vector<size_t> &shape, std::vector<T> &out)
{
    if (is_h5(path)) {
        H5 &L = h5();
        if (!L.load()) throw std::runtime_error("HDF5 input requested but libhdf5 could not be loaded (set DLCO_HDF5_LIB, or pass a directory of .npy files)");
        const H5::hid_t f = L.H5Fopen(path.c_str(), 0, 0);
        if (f < 0) throw std::runtime_error("cannot open " + path);
        if (L.H5Lexists(f, name, 0) <= 0) { L.H5Fclose(f); throw std::runtime_error(path + ": no dataset " + name); }
        const H5::hid_t d = L.H5Dopen2(f, name, 0), s = L.H5Dget_space(d);
        const int nd = L.H5Sget_simple_extent_ndims(s);
        std::vector<unsigned long long> dims(nd > 0 ? nd : 1, 1);
        L.H5Sget_simple_extent_dims(s, dims.data(), nullptr);
        shape.assign(dims.begin(), dims.begin() + (nd > 0 ? nd : 0));
        size_t n = 1;
        for (size_t v : shape) n *= v;
        out.resize(n);
        const int rc = L.H5Dread(d, ElemType<T>::h5t(L), 0, 0, 0, out.data());
        L.H5Sclose(s); L.H5Dclose(d); L.H5Fclose(f);
        if (rc < 0) throw std::runtime_error(path + ": read of " + name + " failed");
    } else {
        std::vector<char> bytes;
        npy_read(path + "/" + name + ".npy", ElemType<T>::npy(), sizeof(T), shape, bytes);
        out.resize(bytes.size() / sizeof(T));
        std::memcpy(out.data(), bytes.data(), bytes.size());
    }
}

struct Writer {
    std::string path;
    H5::hid_t f = -1;
    explicit Writer(const std::string &p) : path(p)
    {
        if (is_h5(path)) {
            H5 &L = h5();
            if (!L.load()) throw std::runtime_error("HDF5 output requested but libhdf5 could not be loaded");
            f = L.H5Fcreate(path.c_str(), 2 /* H5F_ACC_TRUNC */, 0, 0);
            if (f < 0) throw std::runtime_error("cannot create " + path);
        } else {
            mkdir(path.c_str(), 0777);
        }
    }
    template <typename T>
    void write(const char *name, const T *data, size_t rows, size_t cols)
    {
        if (is_h5(path)) {
            H5 &L = h5();
            const unsigned long long dims[2] = {rows, cols};
            const H5::hid_t s = L.H5Screate_simple(2, dims, nullptr);
            const H5::hid_t d = L.H5Dcreate2(f, name, ElemType<T>::h5t(L), s, 0, 0, 0);
            if (d < 0) throw std::runtime_error(path + ": cannot create dataset " + name);
            const int rc = (rows != 0 && cols != 0) ? L.H5Dwrite(d, ElemType<T>::h5t(L), 0, 0, 0, data) : 0;
            L.H5Dclose(d); L.H5Sclose(s);
            if (rc < 0) throw std::runtime_error(path + ": write of dataset " + name + " failed");
        } else {
            npy_write(path + "/" + name + ".npy", data, ElemType<T>::npy(), sizeof(T), {rows, cols});
        }
    }
    void write_f32(const char *name, const float *data, size_t rows, size_t cols) { write<float>(name, data, rows, cols); }
    ~Writer() { if (f >= 0) h5().H5Fclose(f); }
};

// A 2-D dataset of known shape written a block of rows at a time, in row order: what the reference's distance
// generators do with dscreate(rows, cols, type, name, 9, {sChunk, 1}) + one dswrite hyperslab per chunk
// (src/comp-uprjdists.cpp:254-256,289-290,337-339).  HDF5: chunked layout {chunk_r, chunk_c}, deflate level
// `deflate` (0 = none); a run that stops half way leaves the rows written so far.  .npy directory: the header
// carries the final shape and the blocks are appended.
template <typename T>
struct RowStream {
    std::string fpath, dname;
    size_t rows = 0, cols = 0, next = 0;
    bool hdf = false;
    H5::hid_t d = -1;
    FILE *fp = nullptr;
    RowStream(Writer &w, const char *name, size_t rows_, size_t cols_, size_t chunk_r, size_t chunk_c, unsigned deflate)
        : fpath(w.path), dname(name), rows(rows_), cols(cols_), hdf(is_h5(w.path))
    {
        if (hdf) {
            H5 &L = h5();
            typedef H5::hid_t hid_t;
            auto sym = [&](const char *n) { void *q = dlsym(L.h, n); if (!q) throw std::runtime_error(std::string("libhdf5 lacks ") + n); return q; };
            auto H5Pcreate = reinterpret_cast<hid_t (*)(hid_t)>(sym("H5Pcreate"));
            auto H5Pset_chunk = reinterpret_cast<int (*)(hid_t, int, const unsigned long long *)>(sym("H5Pset_chunk"));
            auto H5Pset_deflate = reinterpret_cast<int (*)(hid_t, unsigned)>(sym("H5Pset_deflate"));
            auto H5Pclose = reinterpret_cast<int (*)(hid_t)>(sym("H5Pclose"));
            hid_t *dcpl_cls = reinterpret_cast<hid_t *>(sym("H5P_CLS_DATASET_CREATE_ID_g"));
            const unsigned long long dims[2] = {rows, cols};
            const hid_t sp = L.H5Screate_simple(2, dims, nullptr);
            hid_t pl = 0;
            if (rows > 0 && cols > 0) {
                const unsigned long long chunk[2] = {std::max<size_t>(1, std::min(chunk_r, rows)), std::max<size_t>(1, std::min(chunk_c, cols))};
                pl = H5Pcreate(*dcpl_cls);
                H5Pset_chunk(pl, 2, chunk);
                if (deflate) H5Pset_deflate(pl, deflate);
            }
            d = L.H5Dcreate2(w.f, name, ElemType<T>::h5t(L), sp, 0, pl, 0);
            if (pl) H5Pclose(pl);
            L.H5Sclose(sp);
            if (d < 0) throw std::runtime_error(fpath + ": cannot create dataset " + name);
        } else {
            const std::string f = fpath + "/" + name + ".npy";
            npy_write(f, nullptr, ElemType<T>::npy(), 0, {rows, cols});       // header only (element size 0: no payload)
            fp = std::fopen(f.c_str(), "ab");
            if (!fp) throw std::runtime_error("cannot append to " + f);
        }
    }
    RowStream(const RowStream &) = delete;
    RowStream &operator=(const RowStream &) = delete;
    void write_rows(size_t row0, size_t n, const T *data)
    {
        if (row0 != next || row0 + n > rows) throw std::runtime_error(fpath + ": rows of " + dname + " must be written in order");
        next += n;
        if (n == 0 || cols == 0) return;
        if (hdf) {
            H5 &L = h5();
            auto H5Sselect_hyperslab = reinterpret_cast<int (*)(H5::hid_t, int, const unsigned long long *, const unsigned long long *,
                                                                  const unsigned long long *, const unsigned long long *)>(dlsym(L.h, "H5Sselect_hyperslab"));
            if (!H5Sselect_hyperslab) throw std::runtime_error("libhdf5 lacks H5Sselect_hyperslab");
            const unsigned long long start[2] = {row0, 0}, count[2] = {n, cols};
            const H5::hid_t fs = L.H5Dget_space(d), ms = L.H5Screate_simple(2, count, nullptr);
            int rc = H5Sselect_hyperslab(fs, 0 /* H5S_SELECT_SET */, start, nullptr, count, nullptr);
            if (rc >= 0) rc = L.H5Dwrite(d, ElemType<T>::h5t(L), ms, fs, 0, data);
            L.H5Sclose(ms); L.H5Sclose(fs);
            if (rc < 0) throw std::runtime_error(fpath + ": write of a block of " + dname + " failed");
        } else if (std::fwrite(data, sizeof(T), n * cols, fp) != n * cols) {
            throw std::runtime_error(fpath + ": short write of " + dname);
        }
    }
    ~RowStream()
    {
        if (d >= 0) h5().H5Dclose(d);
        if (fp) std::fclose(fp);
    }
};

// Appends one f32 row to the 2-D dataset `name` of `path`, creating it with unlimited rows, chunk {1, cols}
// and gzip 9 when it does not exist yet (the "w" dataset of pr-learn, src/pr-learn.cpp:391-410).  In the
// .npy-directory form the rows accumulate in `path`/`name`.npy (the file is rewritten).
inline void append_row_f32(const std::string &path, const char *name, const float *row, size_t cols)
{
    if (!is_h5(path)) {
        mkdir(path.c_str(), 0777);
        const std::string f = path + "/" + name + ".npy";
        std::vector<size_t> shape;
        std::vector<char> bytes;
        struct stat st;
        if (::stat(f.c_str(), &st) == 0) npy_read(f, "<f4", 4, shape, bytes);
        if (shape.size() != 2 || shape[1] != cols) { shape = {0, cols}; bytes.clear(); }
        bytes.insert(bytes.end(), reinterpret_cast<const char *>(row), reinterpret_cast<const char *>(row) + cols * 4);
        shape[0]++;
        npy_write(f, bytes.data(), "<f4", 4, shape);
        return;
    }
    H5 &L = h5();
    if (!L.load()) throw std::runtime_error("HDF5 output requested but libhdf5 could not be loaded");
    typedef H5::hid_t hid_t;
    auto sym = [&](const char *n) { void *p = dlsym(L.h, n); if (!p) throw std::runtime_error(std::string("libhdf5 lacks ") + n); return p; };
    auto H5Pcreate = reinterpret_cast<hid_t (*)(hid_t)>(sym("H5Pcreate"));
    auto H5Pset_chunk = reinterpret_cast<int (*)(hid_t, int, const unsigned long long *)>(sym("H5Pset_chunk"));
    auto H5Pset_deflate = reinterpret_cast<int (*)(hid_t, unsigned)>(sym("H5Pset_deflate"));
    auto H5Pclose = reinterpret_cast<int (*)(hid_t)>(sym("H5Pclose"));
    auto H5Dset_extent = reinterpret_cast<int (*)(hid_t, const unsigned long long *)>(sym("H5Dset_extent"));
    auto H5Sselect_hyperslab = reinterpret_cast<int (*)(hid_t, int, const unsigned long long *, const unsigned long long *,
                                                          const unsigned long long *, const unsigned long long *)>(sym("H5Sselect_hyperslab"));
    hid_t *dcpl_cls = reinterpret_cast<hid_t *>(sym("H5P_CLS_DATASET_CREATE_ID_g"));
    struct stat st;
    hid_t f = ::stat(path.c_str(), &st) == 0 ? L.H5Fopen(path.c_str(), 1 /* H5F_ACC_RDWR */, 0) : L.H5Fcreate(path.c_str(), 2, 0, 0);
    if (f < 0) throw std::runtime_error("cannot open " + path + " for writing");
    hid_t d;
    unsigned long long rows = 0;
    if (L.H5Lexists(f, name, 0) > 0) {
        d = L.H5Dopen2(f, name, 0);
        const hid_t s0 = L.H5Dget_space(d);
        unsigned long long dims[2] = {0, 0};
        L.H5Sget_simple_extent_dims(s0, dims, nullptr);
        L.H5Sclose(s0);
        if (dims[1] != cols) { L.H5Dclose(d); L.H5Fclose(f); throw std::runtime_error(path + ": dataset " + name + " has another width"); }
        rows = dims[0];
    } else {
        const unsigned long long dims[2] = {0, cols}, maxd[2] = {~0ULL /* H5S_UNLIMITED */, cols}, chunk[2] = {1, cols};
        const hid_t sp = L.H5Screate_simple(2, dims, maxd), pl = H5Pcreate(*dcpl_cls);
        H5Pset_chunk(pl, 2, chunk);
        H5Pset_deflate(pl, 9);
        d = L.H5Dcreate2(f, name, L.native_float, sp, 0, pl, 0);
        H5Pclose(pl); L.H5Sclose(sp);
        if (d < 0) { L.H5Fclose(f); throw std::runtime_error(path + ": cannot create dataset " + name); }
    }
    const unsigned long long newdims[2] = {rows + 1, cols}, start[2] = {rows, 0}, count[2] = {1, cols};
    int rc = H5Dset_extent(d, newdims);
    const hid_t fs = L.H5Dget_space(d), ms = L.H5Screate_simple(2, count, nullptr);
    if (rc >= 0) rc = H5Sselect_hyperslab(fs, 0 /* H5S_SELECT_SET */, start, nullptr, count, nullptr);
    if (rc >= 0) rc = L.H5Dwrite(d, L.native_float, ms, fs, 0, row);
    L.H5Sclose(ms); L.H5Sclose(fs); L.H5Dclose(d); L.H5Fclose(f);
    if (rc < 0) throw std::runtime_error(path + ": append to dataset " + name + " failed");
}

// GDAL-style progress bar of the reference (src/misc.cpp:45-76)
inline int term_progress(double complete, int last_tick)
{
    int tick = (int)(complete * 40.0);
    if (tick < 0) tick = 0;
    if (tick > 40) tick = 40;
    if (tick < last_tick && last_tick >= 39) last_tick = -1;
    if (tick <= last_tick) return last_tick;
    while (tick > last_tick) {
        last_tick++;
        if (last_tick % 4 == 0) std::printf("%d", (last_tick / 4) * 10);
        else std::printf(".");
    }
    if (tick == 40) std::printf(" - done.\n");
    else std::fflush(stdout);
    return last_tick;
}

}  // namespace dlco_io
