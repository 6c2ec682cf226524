// Prints values of libstdc++'s std::mt19937 and std::uniform_int_distribution<int> so that the C
// restatement in misc3d_oracle.c (orc_mt_next / orc_uniform_int / orc_sample) can be pinned
// against the real standard-library generator the reference uses (include/misc3d/utils.h:73-97).
// Usage: std_rng_check <seed> <count> <mod_size> <range_incl>
//   line 1: <count> raw mt19937 outputs
//   line 2: <count> values of rng() % mod_size
//   line 3: <count> values of uniform_int_distribution<int>(0, range_incl)(rng)   (fresh rng)
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <random>

int main(int argc, char** argv) {
    if (argc < 5) return 2;
    const unsigned long seed = std::strtoul(argv[1], nullptr, 10);
    const int count = std::atoi(argv[2]);
    const size_t mod = std::strtoull(argv[3], nullptr, 10);
    const int range = std::atoi(argv[4]);
    {
        std::mt19937 rng(seed);
        for (int i = 0; i < count; ++i) std::printf("%lu ", (unsigned long)rng());
        std::printf("\n");
    }
    {
        std::mt19937 rng(seed);
        for (int i = 0; i < count; ++i) std::printf("%zu ", (size_t)(rng() % mod));
        std::printf("\n");
    }
    {
        std::mt19937 rng(seed);
        std::uniform_int_distribution<int> dist(0, range);
        for (int i = 0; i < count; ++i) std::printf("%d ", dist(rng));
        std::printf("\n");
    }
    return 0;
}
