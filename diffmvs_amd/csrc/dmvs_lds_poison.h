// DMVS_LDS_POISON(array): nothing on the GPU; under the host emulation (tests/hipemu) the array is filled with NaN patterns at the
// start of every workgroup, so that a kernel reading LDS it never wrote fails its parity test on the CPU as it would with real LDS
// leftovers on the GPU.  (Its own header: dmvs_common.h is part of the source hash roofline.traffic is tied to.)
#pragma once
#ifdef DMVS_HOST_EMULATION
#define DMVS_LDS_POISON(a) hipemu_poison_lds((void*)(a), sizeof(a))
#else
#define DMVS_LDS_POISON(a)
#endif
