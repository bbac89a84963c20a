#!/usr/bin/env python3
"""What the host link gives: page-locked buffers, hipMemcpyAsync up, down and both ways at once on two streams, as one
linear piece and as the 2-D pieces the chunked host path issues (rows = streams, row length = a chunk of frames), for the f32 and int16
shapes of 4096 / 65536 streams x 48 frames.  usage: bus_probe.py [streams] [frames]"""
import ctypes as C
import sys
import time

hip = C.CDLL("libamdhip64.so")
vp, sz = C.c_void_p, C.c_size_t
hip.hipHostMalloc.argtypes = [C.POINTER(vp), sz, C.c_uint]
hip.hipMalloc.argtypes = [C.POINTER(vp), sz]
hip.hipMemcpyAsync.argtypes = [vp, vp, sz, C.c_int, vp]
hip.hipMemcpy2DAsync.argtypes = [vp, sz, vp, sz, sz, sz, C.c_int, vp]
hip.hipStreamCreateWithFlags.argtypes = [C.POINTER(vp), C.c_uint]
hip.hipStreamSynchronize.argtypes = [vp]
H2D, D2H = 1, 2


def chk(e):
    if e:
        raise RuntimeError(f"hip error {e}")


def main():
    S = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    T = int(sys.argv[2]) if len(sys.argv) > 2 else 48
    s_in, s_out = vp(), vp()
    chk(hip.hipStreamCreateWithFlags(C.byref(s_in), 1))
    chk(hip.hipStreamCreateWithFlags(C.byref(s_out), 1))
    for esz, name in ((4, "f32"), (2, "i16")):
        n = S * T * 480 * esz
        hin, hout, din, dout = vp(), vp(), vp(), vp()
        chk(hip.hipHostMalloc(C.byref(hin), n, 0)); chk(hip.hipHostMalloc(C.byref(hout), n, 0))
        chk(hip.hipMalloc(C.byref(din), n)); chk(hip.hipMalloc(C.byref(dout), n))
        C.memset(hin, 1, n)

        def timed(fn, reps=5):
            fn(); hip.hipStreamSynchronize(s_in); hip.hipStreamSynchronize(s_out)
            t0 = time.perf_counter()
            for _ in range(reps):
                fn()
            hip.hipStreamSynchronize(s_in); hip.hipStreamSynchronize(s_out)
            return (time.perf_counter() - t0) / reps

        up = timed(lambda: chk(hip.hipMemcpyAsync(din, hin, n, H2D, s_in)))
        dn = timed(lambda: chk(hip.hipMemcpyAsync(hout, dout, n, D2H, s_out)))
        both = timed(lambda: (chk(hip.hipMemcpyAsync(din, hin, n, H2D, s_in)), chk(hip.hipMemcpyAsync(hout, dout, n, D2H, s_out))))
        print(f"{name} {S} x {T}: {n / 1e6:.0f} MB each way | linear: up {n / up / 1e9:.1f} GB/s, down {n / dn / 1e9:.1f}, both at once {2 * n / both / 1e9:.1f} (sum)")
        pitch = T * 480 * esz
        for chunk in (4, 8, 16, 24, 48):
            w = chunk * 480 * esz
            nch = T // chunk

            def two_d(direction):
                for i in range(nch):
                    off = i * w
                    if direction & 1:
                        chk(hip.hipMemcpy2DAsync(din.value + off, pitch, hin.value + off, pitch, w, S, H2D, s_in))
                    if direction & 2:
                        chk(hip.hipMemcpy2DAsync(hout.value + off, pitch, dout.value + off, pitch, w, S, D2H, s_out))
            u2, d2, b2 = timed(lambda: two_d(1)), timed(lambda: two_d(2)), timed(lambda: two_d(3))
            print(f"    2-D pieces of {chunk:2d} frames ({S} rows x {w} B, {nch} per direction): up {n / u2 / 1e9:.1f}, down {n / d2 / 1e9:.1f}, both {2 * n / b2 / 1e9:.1f} GB/s")
        # frame-major staging: the same bytes as contiguous chunk-sized linear copies (what a [frame][stream] layout would allow)
        for chunk in (8, 16):
            w = S * chunk * 480 * esz
            nch = T // chunk

            def lin(direction):
                for i in range(nch):
                    off = i * w
                    if direction & 1:
                        chk(hip.hipMemcpyAsync(din.value + off, hin.value + off, w, H2D, s_in))
                    if direction & 2:
                        chk(hip.hipMemcpyAsync(hout.value + off, dout.value + off, w, D2H, s_out))
            b2 = timed(lambda: lin(3))
            print(f"    linear pieces of {chunk} frames x all streams ({w / 1e6:.1f} MB): both {2 * n / b2 / 1e9:.1f} GB/s")
        for p in (hin, hout):
            hip.hipHostFree(p)
        for p in (din, dout):
            hip.hipFree(p)


main()
