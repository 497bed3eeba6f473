// Probe (GPU box): what the HIP virtual-memory-management calls return on this driver, step by step.
// Stops at the first failure and never touches memory whose mapping or access call failed.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
static void must(hipError_t e, const char *what)
{
    printf("%-34s -> %d (%s)\n", what, (int)e, hipGetErrorString(e));
    fflush(stdout);
    if (e != hipSuccess)
        exit(3);
}
int main(int argc, char **argv)
{
    const size_t G = (argc > 1 ? strtoull(argv[1], nullptr, 10) : 2) << 20; // slab alignment under test (MiB)
    must(hipSetDevice(0), "hipSetDevice");
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    size_t gmin = 0, grec = 0;
    must(hipMemGetAllocationGranularity(&gmin, &prop, hipMemAllocationGranularityMinimum), "granularity(min)");
    must(hipMemGetAllocationGranularity(&grec, &prop, hipMemAllocationGranularityRecommended), "granularity(rec)");
    printf("granularity min %zu recommended %zu; using %zu\n", gmin, grec, G);
    // equal-size slabs: `count` slabs of `slab` bytes behind one another
    struct Case { size_t slab; int count; };
    for (Case cs : {Case{G, 64}, Case{(size_t)64 << 20, 40}, Case{(size_t)1 << 30, 6}}) {
        const size_t resv = (size_t)16 << 30;
        void *p = nullptr;
        must(hipMemAddressReserve(&p, resv, cs.slab, nullptr, 0), "reserve");
        printf("case: %d slabs of %zu MiB; reserved at %p\n", cs.count, cs.slab >> 20, p);
        size_t mapped = 0;
        std::vector<hipMemGenericAllocationHandle_t> hs;
        for (int i = 0; i < cs.count; ++i) {
            hipMemGenericAllocationHandle_t h;
            hipError_t e = hipMemCreate(&h, cs.slab, &prop, 0);
            if (e == hipSuccess) e = hipMemMap((char *)p + mapped, cs.slab, 0, h, 0);
            if (e != hipSuccess) { printf("  slab %d: create/map -> %d (%s)\n", i, (int)e, hipGetErrorString(e)); return 3; }
            hipMemAccessDesc acc = {};
            acc.location = prop.location;
            acc.flags = hipMemAccessFlagsProtReadWrite;
            e = hipMemSetAccess((char *)p + mapped, cs.slab, &acc, 1);
            if (e != hipSuccess) { printf("  slab %d at +%zu MiB: setaccess -> %d (%s)\n", i, mapped >> 20, (int)e, hipGetErrorString(e)); return 3; }
            hs.push_back(h);
            mapped += cs.slab;
        }
        printf("  all %d slabs mapped and accessible\n", cs.count);
        must(hipMemset(p, 7, mapped), "memset over all slabs");
        must(hipDeviceSynchronize(), "sync");
        unsigned char probe[4] = {0, 0, 0, 0};
        must(hipMemcpy(probe, (char *)p + mapped - 4, 4, hipMemcpyDeviceToHost), "read back the last bytes");
        printf("  last bytes %d %d %d %d\n", probe[0], probe[1], probe[2], probe[3]);
        for (int i = 0; i < cs.count; ++i) {
            must(hipMemUnmap((char *)p + (size_t)i * cs.slab, cs.slab), "unmap");
            must(hipMemRelease(hs[i]), "release");
        }
        must(hipMemAddressFree(p, resv), "addressfree");
    }
    printf("OK\n");
    return 0;
}
