// rccl_dup_probe.hip — can an in-process RCCL communicator be rehearsed on ONE GPU?  (why the multi-device
// one-call path gathers with peer copies and not with ncclCommInitAll + ncclAllGather.)
//   hipcc --offload-arch=gfx950 -o tools/rccl_dup_probe tools/rccl_dup_probe.hip -lrccl
// Prints what ncclCommInitAll says to the device lists {0} and {0, 0}.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <cstdio>

int main()
{
    int n = 0;
    (void)hipGetDeviceCount(&n);
    printf("{\"devices\": %d", n);
    {
        ncclComm_t comm[1];
        const int devs[1] = {0};
        const ncclResult_t r = ncclCommInitAll(comm, 1, devs);
        printf(", \"init_all_[0]\": \"%s\"", ncclGetErrorString(r));
        if (r == ncclSuccess) ncclCommDestroy(comm[0]);
    }
    {
        ncclComm_t comm[2];
        const int devs[2] = {0, 0};
        const ncclResult_t r = ncclCommInitAll(comm, 2, devs);
        printf(", \"init_all_[0,0]\": \"%s\"", ncclGetErrorString(r));
        if (r == ncclSuccess) { ncclCommDestroy(comm[0]); ncclCommDestroy(comm[1]); }
    }
    printf("}\n");
    return 0;
}
