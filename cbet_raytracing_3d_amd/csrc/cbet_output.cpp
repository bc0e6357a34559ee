// cbet_output.cpp -- output stage of the path: the `-D PRINT` text
// rendering the reference's `make test` byte-compares (/root/reference/main.cu:6-22, 353-355;
// Makefile:14-17) and the 27-point node average `edepavg` (main.cu:334-349, dead code there).
#include <cstdio>
#include <cstring>
#include <vector>

#include "cbet_mi355x.h"

namespace cbet {
int fail(int code, const char *fmt, ...);
}

namespace {

// operator<<(ostream&, double) at default precision is printf("%g"): 6 significant digits.
// A small formatter that avoids one printf call per value (12.5 MB of text at 100^3).
struct TextSink {
    FILE *f;
    std::vector<char> buf;
    size_t used = 0;
    long long total = 0;
    explicit TextSink(FILE *fp) : f(fp), buf(1 << 20) {}
    void flush()
    {
        if (used) std::fwrite(buf.data(), 1, used, f);
        total += (long long)used;
        used = 0;
    }
    void put(char c)
    {
        if (used + 1 > buf.size()) flush();
        buf[used++] = c;
    }
    void put(const char *s)
    {
        while (*s) put(*s++);
    }
    void number(double v)
    {
        if (used + 40 > buf.size()) flush();
        used += (size_t)std::snprintf(buf.data() + used, 40, "%g", v);
    }
};

}  // namespace

extern "C" long long cbet_write_text(const double *edep, int d0, int d1, int d2, const char *path)
{
    if (!edep || d0 < 1 || d1 < 1 || d2 < 1) return cbet::fail(CBET_EINVAL, "cbet_write_text: bad array");
    const bool to_stdout = !path || !std::strcmp(path, "-");
    FILE *f = to_stdout ? stdout : std::fopen(path, "wb");
    if (!f) return cbet::fail(CBET_EINVAL, "cbet_write_text: cannot open %s", path);
    TextSink out(f);
    // main.cu:11-22: print(A) = "[" + print(A[0]) + "," + ... + "]" + endl, recursively; a
    // sub-array's own "]\n" therefore precedes the comma that separates it from its sibling.
    out.put('[');
    for (int i = 0; i < d0; ++i) {
        out.put('[');
        for (int j = 0; j < d1; ++j) {
            out.put('[');
            const double *row = edep + ((long)i * d1 + j) * d2;
            for (int k = 0; k < d2; ++k) {
                out.number(row[k]);
                if (k + 1 != d2) out.put(',');
            }
            out.put("]\n");
            if (j + 1 != d1) out.put(',');
        }
        out.put("]\n");
        if (i + 1 != d0) out.put(',');
    }
    out.put("]\n");
    out.flush();
    if (to_stdout) std::fflush(f); else std::fclose(f);
    return out.total;
}

// main.cu:334-349: edepavg[i][j][k] = (sum of the 27 haloed cells edep[i..i+2][j..j+2][k..k+2]) / 27,
// terms added in the order the reference writes them (k-offset outermost, then j, then i).
extern "C" int cbet_edep_average(const double *edep, double *edepavg, int nx, int ny, int nz)
{
    if (!edep || !edepavg || nx < 1 || ny < 1 || nz < 1) return cbet::fail(CBET_EINVAL, "cbet_edep_average: bad array");
    const long sY = nz + 2, sX = (long)(ny + 2) * (nz + 2);
    for (int i = 0; i < nx; ++i)
        for (int j = 0; j < ny; ++j)
            for (int k = 0; k < nz; ++k) {
                double acc = 0.0;
                bool first = true;
                for (int dk = 0; dk < 3; ++dk)
                    for (int dj = 0; dj < 3; ++dj)
                        for (int di = 0; di < 3; ++di) {
                            const double v = edep[(i + di) * sX + (j + dj) * sY + (k + dk)];
                            acc = first ? v : acc + v;
                            first = false;
                        }
                edepavg[((long)i * ny + j) * nz + k] = acc / 27;
            }
    return CBET_OK;
}

// main.cu:321-332 (commented out there): the node coordinate arrays x[i][j][k] = i*dx + xmin, ...
extern "C" int cbet_node_coordinates(const cbet_params *p, double *x, double *y, double *z)
{
    cbet_derived d;
    if (int rc = cbet_derive(p, &d)) return rc;
    if (!x || !y || !z) return cbet::fail(CBET_EINVAL, "cbet_node_coordinates: NULL array");
    for (int i = 0; i < p->nx; ++i)
        for (int j = 0; j < p->ny; ++j)
            for (int k = 0; k < p->nz; ++k) {
                const long o = ((long)i * p->ny + j) * p->nz + k;
                x[o] = i * d.dx + p->xmin;
                y[o] = j * d.dy + p->ymin;
                z[o] = k * d.dz + p->zmin;
            }
    return CBET_OK;
}

// The reference's binary output is save2Hdf5 (main.cu:37-94: /Coordinate_x,y,z and /Edepavg as
// [nx][ny][nz] little-endian fp64), dead code there and dependent on libhdf5, which this image does not
// have.  The same arrays can be written as NumPy .npy files instead (format 1.0: magic, header length,
// a Python-literal header padded to a 64-byte boundary, then the C-order little-endian data).
extern "C" long long cbet_write_npy(const double *data, int ndim, const long *shape, const char *path)
{
    if (!data || !shape || !path || ndim < 1 || ndim > 8) return cbet::fail(CBET_EINVAL, "cbet_write_npy: bad arguments");
    long long count = 1;
    char dims[256];
    size_t used = 0;
    for (int a = 0; a < ndim; ++a) {
        if (shape[a] < 0) return cbet::fail(CBET_EINVAL, "cbet_write_npy: negative extent");
        count *= shape[a];
        used += (size_t)std::snprintf(dims + used, sizeof dims - used, "%ld,%s", shape[a], a + 1 < ndim ? " " : "");
    }
    if (ndim > 1) dims[used - 1] = '\0';   // "(3, 4,)" is legal Python but numpy writes "(3, 4)"
    char header[512];
    int len = std::snprintf(header, sizeof header, "{'descr': '<f8', 'fortran_order': False, 'shape': (%s), }", dims);
    const int unpadded = 10 + len + 1;                       // magic(6) + version(2) + header length(2) + text + '\n'
    const int pad = (64 - unpadded % 64) % 64;
    FILE *f = std::fopen(path, "wb");
    if (!f) return cbet::fail(CBET_EINVAL, "cbet_write_npy: cannot open %s", path);
    const unsigned char magic[8] = {0x93, 'N', 'U', 'M', 'P', 'Y', 1, 0};
    const unsigned hlen = (unsigned)(len + pad + 1);
    const unsigned char hl[2] = {(unsigned char)(hlen & 0xFF), (unsigned char)(hlen >> 8)};
    bool ok = std::fwrite(magic, 1, 8, f) == 8 && std::fwrite(hl, 1, 2, f) == 2 && std::fwrite(header, 1, (size_t)len, f) == (size_t)len;
    for (int i = 0; ok && i < pad; ++i) ok = std::fputc(' ', f) != EOF;
    ok = ok && std::fputc('\n', f) != EOF;
    ok = ok && std::fwrite(data, sizeof(double), (size_t)count, f) == (size_t)count;
    ok = (std::fclose(f) == 0) && ok;
    if (!ok) return cbet::fail(CBET_EINVAL, "cbet_write_npy: short write to %s", path);
    return 10 + (long long)hlen + count * (long long)sizeof(double);
}
