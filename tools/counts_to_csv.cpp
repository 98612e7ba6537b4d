// raw int32 [n][g] (little endian, row-major) -> "index,g0,g1,...\nc0,3,0,...": the integer count CSV the deepImpute CLI is specified for.
//   g++ -O2 -o /tmp/counts_to_csv tools/counts_to_csv.cpp && /tmp/counts_to_csv in.bin n g out.csv
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
int main(int argc, char** argv) {
    if (argc != 5) return 2;
    const long n = atol(argv[2]), g = atol(argv[3]);
    FILE* in = fopen(argv[1], "rb"); FILE* out = fopen(argv[4], "wb");
    if (!in || !out) return 3;
    std::vector<char> buf((size_t)g * 12 + 64);
    std::vector<int32_t> row((size_t)g);
    static char wb[1 << 22]; setvbuf(out, wb, _IOFBF, sizeof wb);
    for (long j = 0; j < g; ++j) fprintf(out, ",g%ld", j);
    fputc('\n', out);
    for (long i = 0; i < n; ++i) {
        if (fread(row.data(), 4, (size_t)g, in) != (size_t)g) return 4;
        char* p = buf.data();
        p += sprintf(p, "c%ld", i);
        for (long j = 0; j < g; ++j) {
            *p++ = ',';
            uint32_t v = (uint32_t)row[(size_t)j];
            char tmp[12]; int k = 0;
            do { tmp[k++] = (char)('0' + v % 10); v /= 10; } while (v);
            while (k) *p++ = tmp[--k];
        }
        *p++ = '\n';
        fwrite(buf.data(), 1, (size_t)(p - buf.data()), out);
    }
    fclose(out); fclose(in);
    return 0;
}
