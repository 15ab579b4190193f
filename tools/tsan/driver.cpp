// tools/tsan_host.sh: the threaded host ingest under ThreadSanitizer, on the device-less HIP of hip_stub.cpp.
// Writes a mixed batch of input files -- plain FASTA, four-line FASTQ, FASTQ whose records span several lines (the host state
// machine's route), .gz of each (one with several members; a corrupt one last) -- and drives the entry points that run the
// stage threads: psk_count_kmers_files (reader + upload | inflate | count), psk_count_kmers_batch (framing pool),
// psk_count_dict_files (prediction's counting), from two contexts on two caller threads at once.
#include "../../include/psk.h"

#include <zlib.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <thread>
#include <vector>

static std::string fasta(std::mt19937 &rng, int n)
{
    std::string s = ">contig one\n";
    for (int i = 0; i < n; i++) {
        s += "ACGT"[rng() & 3];
        if (i % 70 == 69) s += '\n';
    }
    return s + "\n";
}
static std::string fastq(std::mt19937 &rng, int reads, bool wrapped)
{
    std::string s;
    for (int r = 0; r < reads; r++) {
        std::string seq;
        for (int i = 0; i < 120; i++) seq += "ACGT"[rng() & 3];
        s += "@read" + std::to_string(r) + "\n";
        if (wrapped) s += seq.substr(0, 70) + "\n" + seq.substr(70) + "\n+\n" + std::string(70, 'I') + "\n" + std::string(50, 'I') + "\n";
        else s += seq + "\n+\n" + std::string(120, 'I') + "\n";
    }
    return s;
}
static std::string gz(const std::string &t, int level)
{
    z_stream z;
    std::memset(&z, 0, sizeof z);
    deflateInit2(&z, level, Z_DEFLATED, 31, 8, Z_DEFAULT_STRATEGY);
    std::string out(deflateBound(&z, t.size()) + 64, '\0');
    z.next_in = (Bytef *)t.data();
    z.avail_in = (uInt)t.size();
    z.next_out = (Bytef *)&out[0];
    z.avail_out = (uInt)out.size();
    deflate(&z, Z_FINISH);
    out.resize(z.total_out);
    deflateEnd(&z);
    return out;
}
static std::string put(const std::string &dir, const std::string &name, const std::string &bytes)
{
    const std::string p = dir + "/" + name;
    FILE *f = std::fopen(p.c_str(), "wb");
    std::fwrite(bytes.data(), 1, bytes.size(), f);
    std::fclose(f);
    return p;
}

static int run(const std::vector<std::string> &paths, const std::vector<size_t> &sizes, int k, int threads, bool expect_gz_error)
{
    psk_ctx *ctx = nullptr;
    if (psk_init(0, &ctx) != PSK_OK) return 1;
    const int n = (int)paths.size();
    std::vector<const char *> p;
    for (auto &s : paths) p.push_back(s.c_str());
    std::vector<uint64_t> nu((size_t)n), nt((size_t)n);
    int bad = 0;
    for (int rep = 0; rep < 3; rep++) {
        if (psk_begin(ctx, k, n, 0, 0) != PSK_OK) bad = 1;
        const int rc = psk_count_kmers_files(ctx, 0, n, p.data(), sizes.data(), nu.data(), nt.data(), threads, 0, 0, 0, nullptr, nullptr);
        if (expect_gz_error ? rc != PSK_EINVAL : rc != PSK_OK) {
            std::fprintf(stderr, "psk_count_kmers_files: rc %d (%s)\n", rc, psk_last_error(ctx));
            bad = 1;
        }
    }
    std::vector<uint64_t> dict = {1, 5, 77, 1000, 4242};
    std::vector<uint32_t> counts((size_t)n * dict.size());
    if (!expect_gz_error && psk_count_dict_files(ctx, n, p.data(), sizes.data(), k, dict.data(), (uint64_t)dict.size(), counts.data(), threads) != PSK_OK) {
        std::fprintf(stderr, "psk_count_dict_files: %s\n", psk_last_error(ctx));
        bad = 1;
    }
    psk_free(ctx);
    return bad;
}

int main(int argc, char **argv)
{
    const std::string dir = argc > 1 ? argv[1] : "/tmp";
    setenv("PSK_GZ_GROUP_MB", "1", 1);        // runs of ~1 MB of text: a call's .gz samples become several runs, so all three stages overlap
    setenv("PSK_GZ_DEVICE_MIN_MB", "0", 1);   // the .gz files take the device route's host code (the stub's kernels find nothing: zlib on the pool)
    std::mt19937 rng(7);
    std::vector<std::string> paths;
    std::vector<size_t> sizes;
    auto add = [&](const std::string &name, const std::string &bytes) {
        paths.push_back(put(dir, name, bytes));
        sizes.push_back(bytes.size());
    };
    for (int i = 0; i < 6; i++) {
        const std::string fa = fasta(rng, 200000 + 1000 * i), fq = fastq(rng, 1500, false), fw = fastq(rng, 800, true);
        add("s" + std::to_string(i) + ".fasta", fa);
        add("s" + std::to_string(i) + ".fastq", fq);
        add("s" + std::to_string(i) + "_wrapped.fastq", fw);
        add("s" + std::to_string(i) + ".fasta.gz", gz(fa, 6));
        add("s" + std::to_string(i) + ".fastq.gz", gz(fq.substr(0, fq.size() / 2), 1) + gz(fq.substr(fq.size() / 2), 9));
        add("s" + std::to_string(i) + "_wrapped.fastq.gz", gz(fw, 6));
    }
    int bad = 0;
    std::thread other([&] { bad |= run(paths, sizes, 21, 4, false); });   // a second context on another caller thread
    bad |= run(paths, sizes, 13, 8, false);
    other.join();
    // a corrupt .gz among good ones: the error path of the stage threads (first error wins, the others wind down)
    std::string broken = gz(fasta(rng, 50000), 6);
    broken[broken.size() / 2] ^= 0x55;
    broken.resize(broken.size() - 9);
    std::vector<std::string> p2(paths.begin(), paths.begin() + 12);
    std::vector<size_t> s2(sizes.begin(), sizes.begin() + 12);
    p2.push_back(put(dir, "broken.fasta.gz", broken));
    s2.push_back(broken.size());
    bad |= run(p2, s2, 13, 8, true);
    std::printf(bad ? "tsan driver: FAILED\n" : "tsan driver: ok (%zu files, two contexts, three rounds each, error path)\n", paths.size());
    return bad;
}
