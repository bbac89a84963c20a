//! frames/s of `DenoiseState::process_frame` (jneem/nnnoiseless 0.5.1) on the host CPU, in the two shapes SURVEY.md
//! section 8(d) asks for.  NOT compiled or run in this repository's build image (no cargo there).
//!
//!   cargo run --release -- [threads] [streams_per_thread] [frames]
//!
//! (i)  one thread, one stream, the shape of the reference's benches/sin.rs: 100 frames of a 440 Hz full-scale
//!      sine, a fresh state per iteration;
//! (ii) `threads` threads, each advancing `streams_per_thread` independent states by `frames` frames of the
//!      synthetic sine + noise mix (statically partitioned, no shared state).
use nnnoiseless::DenoiseState;
use std::time::Instant;

const FRAME: usize = DenoiseState::FRAME_SIZE;

fn sine_frames(n_frames: usize, freq: f32, amp: f32, noise: f32, seed: u64) -> Vec<f32> {
    let mut s = seed.wrapping_mul(0x9E37_79B9_7F4A_7C15) | 1;
    (0..n_frames * FRAME)
        .map(|i| {
            // xorshift noise, uniform in [-1, 1): the exact distribution does not matter for timing
            s ^= s << 13;
            s ^= s >> 7;
            s ^= s << 17;
            let u = (s >> 40) as f32 / (1u64 << 23) as f32 - 1.0;
            (amp * (2.0 * std::f32::consts::PI * freq * i as f32 / 48_000.0).sin() + noise * u).round()
        })
        .collect()
}

fn run_stream(input: &[f32]) -> f32 {
    let mut st = DenoiseState::new();
    let mut out = [0.0f32; FRAME];
    let mut acc = 0.0;
    for frame in input.chunks_exact(FRAME) {
        acc += st.process_frame(&mut out, frame);
    }
    acc + out[0]
}

fn main() {
    let args: Vec<usize> = std::env::args().skip(1).filter_map(|a| a.parse().ok()).collect();
    let threads = *args.first().unwrap_or(&std::thread::available_parallelism().map(|n| n.get()).unwrap_or(1));
    let per_thread = *args.get(1).unwrap_or(&64);
    let frames = *args.get(2).unwrap_or(&200);

    // (i) benches/sin.rs shape
    let sin = sine_frames(100, 440.0, i16::MAX as f32, 0.0, 1);
    let iters = 200;
    let t0 = Instant::now();
    let mut sink = 0.0;
    for _ in 0..iters {
        sink += run_stream(&sin);
    }
    let dt = t0.elapsed().as_secs_f64();
    println!("single thread, sin.rs shape: {:.0} frames/s ({} x 100 frames, sink {})", (iters * 100) as f64 / dt, iters, sink);

    // (ii) all cores
    let t0 = Instant::now();
    let handles: Vec<_> = (0..threads)
        .map(|t| {
            std::thread::spawn(move || {
                let mut sink = 0.0;
                for s in 0..per_thread {
                    let id = (t * per_thread + s) as u64;
                    let x = sine_frames(frames, 80.0 + (id % 920) as f32, 500.0 + (id % 11_500) as f32, 50.0 + (id % 2_950) as f32, id + 7);
                    sink += run_stream(&x);
                }
                sink
            })
        })
        .collect();
    let sink: f32 = handles.into_iter().map(|h| h.join().unwrap()).sum();
    let dt = t0.elapsed().as_secs_f64();
    let total = (threads * per_thread * frames) as f64;
    println!(
        "{} threads x {} streams x {} frames: {:.0} frames/s total, {:.0} per thread (includes input generation; sink {})",
        threads, per_thread, frames, total / dt, total / dt / threads as f64, sink
    );
}
