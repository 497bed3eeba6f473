//! CPU baseline in the reference's own idiom: one `core::simd` register per 64-lane chunk, ordered
//! `reduce_sum()` per chunk (the reference's `dynamic_simd_cosine!` / `dynamic_simd_euclidean!`,
//! vfs/src/vfs/rank.rs:14-56, accept one register = the whole vector, d <= 64; for d = 768 the chunk sums are
//! added in chunk order).  Build-owned code: same synthetic corpus as bench.py (vfs_amd/synth.py), 262144 rows x
//! 768, Cosine, top-10 by (distance, position); prints one JSON line.  Timing context only -- the parity
//! arithmetic is the strictly sequential sum of oracle/vfs_oracle.c.
#![feature(portable_simd)]
use std::simd::prelude::*;
use std::time::Instant;

const DIM: usize = 768;
const ROWS: usize = 262_144;
const FULL_ROWS: f64 = 1_000_000.0;
const K: usize = 10;
const LANES: usize = 64;

fn splitmix64(mut z: u64) -> u64 {
    z = z.wrapping_add(0x9E37_79B9_7F4A_7C15);
    z = (z ^ (z >> 30)).wrapping_mul(0xBF58_476D_1CE4_E5B9);
    z = (z ^ (z >> 27)).wrapping_mul(0x94D0_49BB_1331_11EB);
    z ^ (z >> 31)
}

fn synth(seed: u64, row0: u64, n: usize) -> Vec<f32> {
    let mut v = Vec::with_capacity(n * DIM);
    for i in 0..n as u64 {
        for j in 0..DIM as u64 {
            let m = (splitmix64(seed.wrapping_add((row0 + i) * DIM as u64 + j)) >> 40) as u32;
            v.push(m as f32 * (1.0 / 8_388_608.0) - 1.0);
        }
    }
    v
}

fn cosine(q: &[f32], x: &[f32]) -> f32 {
    let (mut dot, mut n1, mut n2) = (0.0f32, 0.0f32, 0.0f32);
    for c in 0..DIM / LANES {
        let a = Simd::<f32, LANES>::from_slice(&q[c * LANES..]);
        let b = Simd::<f32, LANES>::from_slice(&x[c * LANES..]);
        dot += (a * b).reduce_sum();
        n1 += (a * a).reduce_sum();
        n2 += (b * b).reduce_sum();
    }
    1.0 - dot / (n1.sqrt() * n2.sqrt())
}

fn main() {
    let corpus = synth(0x5EED_0001, 0, ROWS);
    let queries = synth(0x5EED_0002, 0, 64);
    let t0 = Instant::now();
    let mut done = 0usize;
    let mut sink = 0.0f32;
    for qi in 0..64 {
        let q = &queries[qi * DIM..(qi + 1) * DIM];
        let mut best: Vec<(f32, usize)> = Vec::with_capacity(K + 1);
        for r in 0..ROWS {
            let d = cosine(q, &corpus[r * DIM..(r + 1) * DIM]);
            if best.len() < K || d < best[best.len() - 1].0 {
                let pos = best.partition_point(|e| e.0 <= d);
                best.insert(pos, (d, r));
                best.truncate(K);
            }
        }
        sink += best[0].0;
        done += 1;
        if t0.elapsed().as_secs_f64() > 8.0 {
            break;
        }
    }
    let dt = t0.elapsed().as_secs_f64();
    let qps = (ROWS * done) as f64 / dt / FULL_ROWS;
    println!(
        "{{\"value\": {:.4}, \"unit\": \"queries/s\", \"cores\": 1, \"kind\": \"rust nightly core::simd, 64-lane chunks (build-owned restatement of rank.rs:36-56 for d = 768)\", \"sample\": \"{} queries x {} rows in {:.1} s, scaled to 1M rows\", \"checksum\": {}}}",
        qps, done, ROWS, dt, sink
    );
}
