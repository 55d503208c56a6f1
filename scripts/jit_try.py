"""Dev helper: compile a source string with hiprtc here (no GPU needed) and print the log."""
import ctypes as C, sys
rtc = C.CDLL("/opt/rocm/lib/libhiprtc.so")
def compile_src(src, opts=("--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off")):
    prog = C.c_void_p()
    rc = rtc.hiprtcCreateProgram(C.byref(prog), src.encode(), b"t.hip", 0, None, None)
    assert rc == 0, rc
    arr = (C.c_char_p * len(opts))(*[o.encode() for o in opts])
    rc = rtc.hiprtcCompileProgram(prog, len(opts), arr)
    n = C.c_size_t()
    rtc.hiprtcGetProgramLogSize(prog, C.byref(n))
    buf = C.create_string_buffer(n.value + 1)
    rtc.hiprtcGetProgramLog(prog, buf)
    return rc, buf.value.decode(errors="replace")
if __name__ == "__main__":
    rc, log = compile_src(open(sys.argv[1]).read())
    print("rc", rc); print(log[:6000])
