#!/usr/bin/env python3
"""tools/device_caches.py -- what the HIP runtime reports about the cache hierarchy of device 0 (L2 bytes, XCC count, CUs):
the quantities the column-band policy of the stateless call is derived from (csrc/mspmv_api.hip: band_passes_for)."""
import ctypes
hip = ctypes.CDLL("libamdhip64.so")
# enum values from /opt/rocm/include/hip/hip_runtime_api.h (hipDeviceAttribute_t); looked up by name at build time in the library,
# here only printed for the record
import re
hdr = open("/opt/rocm/include/hip/hip_runtime_api.h").read()
body = hdr[hdr.index("typedef enum hipDeviceAttribute_t"):]
body = body[:body.index("} hipDeviceAttribute_t")]
names, val = {}, -1
for line in body.splitlines():
    m = re.match(r"\s*(hipDeviceAttribute\w+)\s*(?:=\s*(\w+))?\s*,?", line)
    if not m: continue
    if m.group(2):
        try: val = int(m.group(2), 0)
        except ValueError: val = names.get(m.group(2), val)
    else: val += 1
    names[m.group(1)] = val
for n in ("hipDeviceAttributeL2CacheSize", "hipDeviceAttributeNumberOfXccs", "hipDeviceAttributeMultiprocessorCount",
          "hipDeviceAttributeMaxSharedMemoryPerMultiprocessor", "hipDeviceAttributeClockRate", "hipDeviceAttributeMemoryClockRate",
          "hipDeviceAttributeMemoryBusWidth", "hipDeviceAttributePersistingL2CacheMaxSize"):
    v = ctypes.c_int(-1)
    rc = hip.hipDeviceGetAttribute(ctypes.byref(v), names[n], 0)
    print(f"{n} (= {names[n]}): rc {rc} value {v.value}")
