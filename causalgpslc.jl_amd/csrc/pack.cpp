// Posterior pack reader / writer behind the C ABI (include/gpslc_hip.h "posterior pack"): the flat
// little-endian replacement of Julia's Serialization of a GPSLCObject for the prediction path
// (src/io.jl:14-34; contents = what extractParameters, src/utils.jl:92-124, returns per retained sample).
// Host-only, no GPU call; gfx950 hosts are little-endian x86-64, so doubles are written as they sit in memory.
#include "../../include/gpslc_hip.h"

#include <cstdio>
#include <cstring>

namespace {

const char kMagic[8] = {'G', 'P', 'S', 'L', 'C', 'P', 'K', '1'};

struct File {
    FILE* f = nullptr;
    ~File() { if (f) fclose(f); }
};

bool header_ok(const gpslc_pack_header& h) {
    return h.n >= 1 && h.nX >= 0 && h.nU >= 0 && h.S >= 0 && h.nX + h.nU <= 32 && (h.binary_t == 0 || h.binary_t == 1);
}

int read_header(FILE* f, gpslc_pack_header* h) {
    char magic[8];
    if (fread(magic, 1, 8, f) != 8 || memcmp(magic, kMagic, 8) != 0) return GPSLC_ERR_FORMAT;
    int64_t dims[6];
    if (fread(dims, 8, 6, f) != 6) return GPSLC_ERR_FORMAT;
    h->n = dims[0]; h->nX = dims[1]; h->nU = dims[2]; h->S = dims[3]; h->binary_t = dims[4]; h->reserved = dims[5];
    if (fread(h->hyper, 8, 7, f) != 7) return GPSLC_ERR_FORMAT;
    return header_ok(*h) ? GPSLC_OK : GPSLC_ERR_FORMAT;
}

// read `count` doubles at the current position into dst (or skip them when dst is NULL)
int take(FILE* f, double* dst, int64_t count) {
    if (count <= 0) return GPSLC_OK;
    if (!dst) return fseek(f, (long)(count * 8), SEEK_CUR) == 0 ? GPSLC_OK : GPSLC_ERR_FORMAT;
    return fread(dst, 8, (size_t)count, f) == (size_t)count ? GPSLC_OK : GPSLC_ERR_FORMAT;
}

// the samples [s0, s1) of an array whose last axis is the sample index and whose leading block is `per` doubles
int take_samples(FILE* f, double* dst, int64_t per, int64_t S, int64_t s0, int64_t s1) {
    if (per <= 0 || S <= 0) return GPSLC_OK;
    int rc = take(f, nullptr, per * s0);
    if (rc) return rc;
    rc = take(f, dst, per * (s1 - s0));
    if (rc) return rc;
    return take(f, nullptr, per * (S - s1));
}

}  // namespace

extern "C" {

int gpslc_pack_save(const char* path, const gpslc_pack_header* h, const double* X, const double* T, const double* Y,
                    const double* U, const double* uyLS, const double* xyLS, const double* tyLS,
                    const double* yNoise, const double* yScale) {
    if (!path) return -1;
    if (!h || !header_ok(*h)) return -2;
    if (h->nX > 0 && !X) return -3;
    if (!T) return -4;
    if (!Y) return -5;
    if (h->nU > 0 && h->S > 0 && (!U || !uyLS)) return -6;
    if (h->nX > 0 && h->S > 0 && !xyLS) return -8;
    if (h->S > 0 && (!tyLS || !yNoise || !yScale)) return -9;
    File fl;
    fl.f = fopen(path, "wb");
    if (!fl.f) return GPSLC_ERR_IO;
    const int64_t dims[6] = {h->n, h->nX, h->nU, h->S, h->binary_t, 0};
    bool ok = fwrite(kMagic, 1, 8, fl.f) == 8 && fwrite(dims, 8, 6, fl.f) == 6 && fwrite(h->hyper, 8, 7, fl.f) == 7;
    auto put = [&](const double* p, int64_t count) {
        if (ok && count > 0) ok = fwrite(p, 8, (size_t)count, fl.f) == (size_t)count;
    };
    put(X, h->n * h->nX);
    put(T, h->n);
    put(Y, h->n);
    put(U, h->n * h->nU * h->S);
    put(uyLS, h->nU * h->S);
    put(xyLS, h->nX * h->S);
    put(tyLS, h->S);
    put(yNoise, h->S);
    put(yScale, h->S);
    if (ok) ok = fflush(fl.f) == 0;
    return ok ? GPSLC_OK : GPSLC_ERR_IO;
}

int gpslc_pack_read_header(const char* path, gpslc_pack_header* h) {
    if (!path) return -1;
    if (!h) return -2;
    File fl;
    fl.f = fopen(path, "rb");
    if (!fl.f) return GPSLC_ERR_IO;
    int rc = read_header(fl.f, h);
    if (rc) return rc;
    // total length must match the header exactly (truncated / trailing bytes are format errors)
    const int64_t doubles = h->n * h->nX + 2 * h->n + h->n * h->nU * h->S + (h->nU + h->nX + 3) * h->S;
    if (fseek(fl.f, 0, SEEK_END) != 0) return GPSLC_ERR_IO;
    const long end = ftell(fl.f);
    return end == (long)(8 + 48 + 56 + 8 * doubles) ? GPSLC_OK : GPSLC_ERR_FORMAT;
}

int gpslc_pack_load(const char* path, int64_t s0, int64_t s1, double* X, double* T, double* Y, double* U,
                    double* uyLS, double* xyLS, double* tyLS, double* yNoise, double* yScale) {
    gpslc_pack_header h;
    int rc = gpslc_pack_read_header(path, &h);
    if (rc) return rc;
    if (s0 < 0 || s0 > s1) return -2;
    if (s1 > h.S) return -3;
    File fl;
    fl.f = fopen(path, "rb");
    if (!fl.f) return GPSLC_ERR_IO;
    if (fseek(fl.f, 8 + 48 + 56, SEEK_SET) != 0) return GPSLC_ERR_IO;
    if ((rc = take(fl.f, X, h.n * h.nX))) return rc;
    if ((rc = take(fl.f, T, h.n))) return rc;
    if ((rc = take(fl.f, Y, h.n))) return rc;
    if ((rc = take_samples(fl.f, U, h.n * h.nU, h.S, s0, s1))) return rc;
    if ((rc = take_samples(fl.f, uyLS, h.nU, h.S, s0, s1))) return rc;
    if ((rc = take_samples(fl.f, xyLS, h.nX, h.S, s0, s1))) return rc;
    if ((rc = take_samples(fl.f, tyLS, 1, h.S, s0, s1))) return rc;
    if ((rc = take_samples(fl.f, yNoise, 1, h.S, s0, s1))) return rc;
    return take_samples(fl.f, yScale, 1, h.S, s0, s1);
}

}  // extern "C"
