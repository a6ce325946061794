// orz_cli.cpp -- the `orz` command line on top of liborz_hip.so.
// Same grammar as the reference binary (/root/reference/src/main.rs:21-52,97-113):
//   orz encode [-s|--silent] [-l|--level 0..2 (default 2)] [source] [target]
//   orz decode [-s|--silent] [source] [target]
// source/target default to stdin/stdout.  Additive flags: --device N (HIP device of the encoder);
//   encode --member-size BYTES [--jobs J]: cut the input into independent members (complete orz streams,
//   concatenated) and encode J of them concurrently on the GPU;  decode --members: decode every stream of
//   such a concatenation (the reference, and the default here, stop after the first stream);
//   encode --verify: every block's bytes go through the library's own decoder and must reproduce the input before they are
//   written (ORZ_VERIFY=decode); a failed check fails the encode;
//   decode --members --gpu [--device N]: decode the members on the GPU, one member per wavefront.
// Progress lines mirror SimpleProgressLogger (src/progress.rs:48-98) on stderr.
#include <cerrno>
#include <chrono>
#include <cstdio>
#include <sys/stat.h>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/orz_hip.h"

namespace {
struct Io {
    FILE* in;
    FILE* out;
};
ssize_t rd(void* c, uint8_t* buf, size_t cap) {
    Io* io = (Io*)c;
    size_t n = fread(buf, 1, cap, io->in);
    if (n == 0 && ferror(io->in)) return -1;
    return (ssize_t)n;
}
int wr(void* c, const uint8_t* buf, size_t n) { return fwrite(buf, 1, n, ((Io*)c)->out) == n ? 0 : -1; }
struct Prog {
    bool encode;
    std::chrono::steady_clock::time_point t0;
};
void progress(void* c, int fin, size_t a, size_t b) {
    Prog* p = (Prog*)c;
    double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - p->t0).count();
    size_t src = p->encode ? a : b, dst = p->encode ? b : a;  // decode reports (compressed in, raw out)
    double mbs = s > 0 ? src / s / 1e6 : 0;
    if (fin)
        fprintf(stderr, "%s: %zu bytes => %zu bytes, ratio %.3f, %.3f MB/s, %.3f s\n", p->encode ? "encode" : "decode",
                p->encode ? a : a, p->encode ? b : b, src ? (double)(p->encode ? dst : a) / (double)(p->encode ? src : b) : 0.0, mbs, s);
    else
        fprintf(stderr, "%s: %zu bytes => %zu bytes, %.3f MB/s\n", p->encode ? "encode" : "decode", a, b, mbs);
}
int usage() {
    fprintf(stderr,
            "an optimized ROLZ data compressor (MI355X encoder)\n\nUsage: orz <COMMAND>\n\nCommands:\n"
            "  encode  Encode   [-s|--silent] [-l|--level <0..2>] [--mode fast|exact] [--backend hip] [--device N]\n"
            "                   [--member-size BYTES [--jobs J] [--gpus N]] [--verify] [source] [target]\n"
            "  decode  Decode   [-s|--silent] [source] [target]\n");
    return 2;
}
}  // namespace

int main(int argc, char** argv) {
    setenv("GPU_MAX_HW_QUEUES", "16", 0);  // (before the HIP runtime starts: see orz_capi.hip)
    if (argc < 2) return usage();
    const std::string cmd = argv[1];
    if (cmd != "encode" && cmd != "decode") return usage();
    bool silent = false;
    long level = 2, device = 0, jobs = 4, gpus = 1;
    long long member_size = 0;
    bool all_members = false, gpu_decode = false;
    std::vector<std::string> pos;
    for (int i = 2; i < argc; i++) {
        std::string a = argv[i];
        if (a == "-s" || a == "--silent") silent = true;
        else if (cmd == "encode" && (a == "-l" || a == "--level")) { if (++i >= argc) return usage(); level = strtol(argv[i], nullptr, 10); }
        else if (cmd == "encode" && a.rfind("--level=", 0) == 0) level = strtol(a.c_str() + 8, nullptr, 10);
        else if (cmd == "encode" && a.rfind("-l", 0) == 0 && a.size() > 2) level = strtol(a.c_str() + 2, nullptr, 10);
        else if (cmd == "encode" && a == "--device") { if (++i >= argc) return usage(); device = strtol(argv[i], nullptr, 10); }
        else if (cmd == "encode" && a == "--member-size") { if (++i >= argc) return usage(); member_size = strtoll(argv[i], nullptr, 10); }
        else if (cmd == "encode" && a == "--jobs") { if (++i >= argc) return usage(); jobs = strtol(argv[i], nullptr, 10); }
        else if (cmd == "encode" && a == "--gpus") { if (++i >= argc) return usage(); gpus = strtol(argv[i], nullptr, 10); }
        else if (cmd == "encode" && (a == "--backend" || a.rfind("--backend=", 0) == 0)) {
            std::string v = a == "--backend" ? (++i < argc ? argv[i] : "") : a.substr(10);
            if (v != "hip") { fprintf(stderr, "Error: \"backend %s: this build has the HIP backend only (there is no CPU fallback)\"\n", v.c_str()); return 1; }
        }
        else if (cmd == "encode" && (a == "--mode" || a.rfind("--mode=", 0) == 0)) {
            std::string v = a == "--mode" ? (++i < argc ? argv[i] : "") : a.substr(7);
            if (v != "fast" && v != "exact") return usage();
            setenv("ORZ_MODE", v.c_str(), 1);  // (read by the library when an encoder is created)
        }
        else if (cmd == "encode" && a == "--verify") setenv("ORZ_VERIFY", "decode", 1);  // round trip through the library's own decoder
        else if (cmd == "decode" && a == "--members") all_members = true;
        else if (cmd == "decode" && a == "--gpu") gpu_decode = true;
        else if (cmd == "decode" && a == "--device") { if (++i >= argc) return usage(); device = strtol(argv[i], nullptr, 10); }
        else if (a.size() > 1 && a[0] == '-' ) return usage();
        else pos.push_back(a);
    }
    if (pos.size() > 2) return usage();
    Io io{stdin, stdout};
    if (pos.size() >= 1 && !(io.in = fopen(pos[0].c_str(), "rb"))) { fprintf(stderr, "Error: %s: %s\n", pos[0].c_str(), strerror(errno)); return 1; }
    if (pos.size() >= 2 && !(io.out = fopen(pos[1].c_str(), "wb"))) { fprintf(stderr, "Error: %s: %s\n", pos[1].c_str(), strerror(errno)); return 1; }
    Prog pg{cmd == "encode", std::chrono::steady_clock::now()};
    int rc;
    if (cmd == "encode") {
        orz_lzcfg cfg;
        if (orz_lzcfg_from_level((int)level, &cfg) != ORZ_OK) {  // src/main.rs:101
            fprintf(stderr, "Error: \"invalid level: %ld\"\n", level);
            return 1;
        }
        if (member_size > 0) {
            if (gpus < 1 || gpus > 64) return usage();
            std::vector<int> devs;
            for (long g = 0; g < gpus; g++) devs.push_back((int)(device + g));  // --gpus N: devices device .. device+N-1
            orz_members* m = orz_members_new_multi(devs.data(), (int)devs.size(), &cfg, (int)jobs);
            if (!m) { fprintf(stderr, "Error: \"encoding failed: %s\"\n", orz_last_error()); return 1; }
            std::vector<uint8_t> buf((size_t)member_size * (size_t)jobs * (size_t)gpus);
            size_t in_total = 0, out_total = 0;
            bool any = false;
            rc = ORZ_OK;
            for (;;) {
                size_t got = fread(buf.data(), 1, buf.size(), io.in);
                if (got == 0 && any) break;
                uint8_t* out = nullptr;
                size_t out_len = 0;
                rc = orz_members_encode(m, buf.data(), got, 0, (size_t)member_size, &out, &out_len, nullptr);
                if (rc != ORZ_OK) break;
                if (fwrite(out, 1, out_len, io.out) != out_len) rc = ORZ_EIO;
                orz_free(out);
                if (rc != ORZ_OK) break;
                any = true;
                in_total += got;
                out_total += out_len;
                if (!silent) progress(&pg, 0, in_total, out_total);
                if (got < buf.size()) break;
            }
            orz_members_free(m);
            if (rc == ORZ_OK && !silent) progress(&pg, 1, in_total, out_total);
        } else {
            rc = orz_encode(rd, &io, wr, &io, &cfg, silent ? nullptr : progress, &pg, (int)device);
        }
        if (rc != ORZ_OK) {
            fprintf(stderr, "Error: \"encoding failed: %s\"\n", orz_last_error());
            // the streaming entry point hands blocks on as they are finished: what a failed encode left behind is a truncated
            // stream, and it goes -- a named target only, never stdout
            // -- and only a REGULAR file (`orz encode in /dev/null` run as root must not unlink the device node, nor
            // a FIFO; what the open stream refers to is asked, not the name)
            if (pos.size() >= 2) {
                struct stat st;
                const bool regular = fstat(fileno(io.out), &st) == 0 && S_ISREG(st.st_mode);
                fclose(io.out);
                io.out = nullptr;
                if (regular && remove(pos[1].c_str()) == 0) fprintf(stderr, "(the incomplete output %s was removed)\n", pos[1].c_str());
            }
            return 1;
        }
    } else if (gpu_decode) {
        std::vector<uint8_t> in;
        uint8_t tmp[1 << 16];
        for (size_t got; (got = fread(tmp, 1, sizeof tmp, io.in)) > 0;) in.insert(in.end(), tmp, tmp + got);
        uint8_t* out = nullptr;
        size_t out_len = 0, nm = 0;
        rc = orz_decode_members_device((int)device, in.data(), in.size(), &out, &out_len, &nm, nullptr);
        if (rc == ORZ_OK && fwrite(out, 1, out_len, io.out) != out_len) rc = ORZ_EIO;
        if (out) orz_free(out);
        if (rc != ORZ_OK) { fprintf(stderr, "Error: \"decoding failed: %s\"\n", orz_last_error()); return 1; }
        if (!silent) progress(&pg, 1, in.size(), out_len);
    } else {
        for (;;) {
            rc = orz_decode(rd, &io, wr, &io, silent ? nullptr : progress, &pg);
            if (rc != ORZ_OK || !all_members) break;
            int c = fgetc(io.in);  // another member behind this stream's EOF chunk?
            if (c == EOF) break;
            ungetc(c, io.in);
        }
        if (rc != ORZ_OK) { fprintf(stderr, "Error: \"decoding failed: %s\"\n", orz_last_error()); return 1; }
    }
    if (fflush(io.out) != 0) { fprintf(stderr, "Error: write failed\n"); return 1; }
    return 0;
}
