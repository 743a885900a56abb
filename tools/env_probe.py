"""Does the HIP runtime's initialisation change the process environment?  (getenv on one thread is only safe
while no other thread calls setenv/putenv; this library's workers therefore read no environment.)  Walks the C-level
`environ` array before and after the first HIP call of a fresh process and reports what moved or appeared."""
import ctypes as C

libc = C.CDLL(None)


def snapshot():
    env = C.POINTER(C.c_char_p).in_dll(libc, "environ")
    base = C.cast(env, C.c_void_p).value
    out, i = [], 0
    while env[i]:
        out.append(env[i].decode(errors="replace"))
        i += 1
    return base, out


before_base, before = snapshot()
hip = C.CDLL("libamdhip64.so")
rc = hip.hipInit(0)
n = C.c_int(0)
hip.hipGetDeviceCount(C.byref(n))
rc2 = hip.hipFree(None)
after_base, after = snapshot()
print(f"hipInit -> {rc}, devices {n.value}, hipFree(nullptr) -> {rc2}")
print(f"environ array: {before_base:#x} -> {after_base:#x} ({'MOVED' if before_base != after_base else 'same address'}), "
      f"{len(before)} -> {len(after)} entries")
b = dict(e.split("=", 1) for e in before if "=" in e)
a = dict(e.split("=", 1) for e in after if "=" in e)
for k in sorted(set(a) - set(b)):
    print(f"  added by the runtime's initialisation: {k}={a[k][:60]}")
for k in sorted(k for k in a if k in b and a[k] != b[k]):
    print(f"  changed by the runtime's initialisation: {k}")
for k in sorted(set(b) - set(a)):
    print(f"  removed: {k}")
