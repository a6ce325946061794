// race_main.cpp -- race check of the encoder's kernels (TEST INFRASTRUCTURE ONLY).
//
// The host emulation (tests/emu) with a launch's wavefronts spread over ORZ_EMU_THREADS host threads, built with
// -fsanitize=thread: every pair of accesses to one location by threads of DIFFERENT wavefronts of one launch, at least one a
// write and not both atomic, is a report -- independent of how this run happened to be scheduled.  The defect of round 3
// (FastWordCheck read ty[i + 1] while the neighbouring wavefront rewrote it) is such a pair; the sequential emulation could
// not see it.  Launches are separated by thread joins, so accesses of different launches are ordered, as on a HIP stream.
//   tests/race/run.sh [file | text:<bytes> | mixed:<bytes>] ...     (see there)
#include "../emu/emu_backend.cpp"

#include <fstream>
#include <iterator>

int main(int argc, char** argv) {
    if (argc < 3) { std::fprintf(stderr, "usage: race_main <fast|exact> <input file>\n"); return 2; }
    std::ifstream in(argv[2], std::ios::binary);
    std::vector<uint8_t> data((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
    uint8_t* dst = nullptr;
    size_t n = 0;
    int rc;
    if (std::string(argv[1]) == "fast") rc = emu_encode_fast(data.data(), data.size(), 15, 9, 6, 0, 0, &dst, &n, nullptr);
    else rc = emu_encode(data.data(), data.size(), 15, 9, 6, 62, 256, 1, &dst, &n, nullptr);
    std::fprintf(stderr, "race_main: %s mode, %zu bytes -> %zu bytes, rc %d\n", argv[1], data.size(), n, rc);
    if (argc > 3 && rc == 0) { std::ofstream out(argv[3], std::ios::binary); out.write((const char*)dst, (std::streamsize)n); }
    return rc;
}
