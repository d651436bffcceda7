// UNCOMPILED SOURCE (no Rust toolchain in the build image). An example for the REFERENCE crate -- its CPU backend, not simd_hip --
// that turns the committed input file of this repository into golden vectors of the real block-aligner:
//
//   cp <this repo>/rust/examples/dump_golden.rs <block-aligner>/examples/
//   cd <block-aligner>
//   cargo run --release --features simd_avx2 --example dump_golden -- <this repo>/tests/golden/crate_golden_input.tsv <this repo>/tests/golden/crate_golden.tsv
//
// tests/test_crate_golden.py picks crate_golden.tsv up when it exists and compares the oracle (CPU test, every line) and the HIP path
// (-m gpu, every line) with it: score, end indices, CIGAR, Trace::blocks(). No dependency beyond the crate itself.
//
// Input: one case per line, tab-separated (tests/golden/make_crate_golden_input.py writes it):
//   id  kind(aa|nuc|bytes|profile)  matrix  gap_open  gap_extend  min_size  max_size  x_drop  mode  query  reference
//   kind "profile" (Block::align_profile): matrix = "pssm" -- row i of the profile is the BLOSUM62 row of reference[i] over the 20 standard residues,
//   as examples/pssm_bench.rs builds one --, gap_open = one "gap_open_C/gap_close_C/gap_open_R" triple per position 0 ..= len, comma-separated,
//   gap_extend = the profile's, reference = the consensus the rows come from.
//   matrix: a static of scores.rs (BLOSUM62, NW1, BYTES1, ...) or "simple:<match>:<mismatch>"; mode: '+'-joined subset of
//   trace,x_drop,local_start,free_query_start_gaps,free_query_end_gaps ("-" = none); sequences are printable ASCII ("-" = empty).
// Output: one line per case:
//   id  score  query_idx  reference_idx  cigar(or "-" without trace; M/I/D ops)  cigar_eq(or "-"; =/X/I/D)  blocks("row,col,width,height;..." or "-")
use block_aligner::cigar::*;
use block_aligner::scan_block::*;
use block_aligner::scores::*;

use std::env;
use std::fs;
use std::io::Write;

struct Case<'a> { gaps: Gaps, min: usize, max: usize, x_drop: i32, q: &'a [u8], r: &'a [u8] }

fn blocks_str(v: &[Rectangle]) -> String {
    if v.is_empty() { return "-".to_string(); }
    v.iter().map(|b| format!("{},{},{},{}", b.row, b.col, b.width, b.height)).collect::<Vec<_>>().join(";")
}

// One alignment with the five mode bits as const parameters; TRACE also yields both CIGAR forms from the end position and blocks().
fn run<M: Matrix, const TRACE: bool, const X_DROP: bool, const LOCAL_START: bool, const FQS: bool, const FQE: bool>(m: &M, c: &Case) -> String {
    let qp = PaddedBytes::from_bytes::<M>(c.q, c.max);
    let rp = PaddedBytes::from_bytes::<M>(c.r, c.max);
    let mut b = Block::<TRACE, X_DROP, LOCAL_START, FQS, FQE>::new(c.q.len(), c.r.len(), c.max);
    b.align(&qp, &rp, m, c.gaps, c.min..=c.max, c.x_drop);
    let res = b.res();
    if TRACE {
        let mut cg = Cigar::new(c.q.len(), c.r.len());
        b.trace().cigar(res.query_idx, res.reference_idx, &mut cg);
        let s1 = cg.to_string();
        b.trace().cigar_eq(&qp, &rp, res.query_idx, res.reference_idx, &mut cg);
        let s2 = cg.to_string();
        let e = |s: String| if s.is_empty() { "-".to_string() } else { s };
        format!("{}\t{}\t{}\t{}\t{}\t{}", res.score, res.query_idx, res.reference_idx, e(s1), e(s2), blocks_str(&b.trace().blocks()))
    } else {
        format!("{}\t{}\t{}\t-\t-\t-", res.score, res.query_idx, res.reference_idx)
    }
}

// Sequence-to-profile: the same with Block::align_profile (scan_block.rs:942-968); no cigar_eq (there is no second sequence) and no blocks().
fn run_profile<const TRACE: bool, const X_DROP: bool, const LOCAL_START: bool, const FQS: bool, const FQE: bool>(p: &AAProfile, c: &Case) -> String {
    let qp = PaddedBytes::from_bytes::<AAMatrix>(c.q, c.max);
    let mut b = Block::<TRACE, X_DROP, LOCAL_START, FQS, FQE>::new(c.q.len(), p.len(), c.max);
    b.align_profile(&qp, p, c.min..=c.max, c.x_drop);
    let res = b.res();
    if TRACE {
        let mut cg = Cigar::new(c.q.len(), p.len());
        b.trace().cigar(res.query_idx, res.reference_idx, &mut cg);
        let s1 = cg.to_string();
        format!("{}\t{}\t{}\t{}\t-\t{}", res.score, res.query_idx, res.reference_idx, if s1.is_empty() { "-".to_string() } else { s1 }, blocks_str(&b.trace().blocks()))
    } else {
        format!("{}\t{}\t{}\t-\t-\t-", res.score, res.query_idx, res.reference_idx)
    }
}

fn dispatch_profile(p: &AAProfile, mode: &str, c: &Case) -> String {
    let has = |k: &str| mode.split('+').any(|x| x == k);
    let (t, x, l, s, e) = (has("trace"), has("x_drop"), has("local_start"), has("free_query_start_gaps"), has("free_query_end_gaps"));
    macro_rules! go { ($t:literal, $x:literal, $l:literal, $s:literal, $e:literal) => { if (t, x, l, s, e) == ($t, $x, $l, $s, $e) { return run_profile::<$t, $x, $l, $s, $e>(p, c); } } }
    macro_rules! both { ($x:literal, $l:literal, $s:literal, $e:literal) => { go!(false, $x, $l, $s, $e); go!(true, $x, $l, $s, $e); } }
    both!(false, false, false, false); both!(true, false, false, false);
    both!(false, true, false, false); both!(true, true, false, false);
    both!(false, false, true, false); both!(true, false, true, false);
    both!(false, false, false, true); both!(false, true, false, true); both!(false, false, true, true);
    panic!("mode {:?} is not one Block::align_profile accepts", mode);
}

// a PSSM as examples/pssm_bench.rs:64-84 builds one: row i = BLOSUM62 row of consensus[i]; per-position gap costs from the triples
fn build_profile(cons: &[u8], triples: &str, gap_extend: i8, max: usize) -> AAProfile {
    let mut p = AAProfile::new(cons.len(), max, gap_extend);
    for (i, &c) in cons.iter().enumerate() {
        for &b in b"ACDEFGHIKLMNPQRSTVWY" { p.set(i + 1, b, BLOSUM62.get(c, b)); }
    }
    for (i, t) in triples.split(',').enumerate() {
        let v: Vec<i8> = t.split('/').map(|x| x.parse().unwrap()).collect();
        p.set_gap_open_C(i, v[0]); p.set_gap_close_C(i, v[1]); p.set_gap_open_R(i, v[2]);
    }
    p
}

// the combinations Block::align accepts (scan_block.rs:860-862: not LOCAL_START with FREE_QUERY_START_GAPS, not X_DROP with FREE_QUERY_END_GAPS)
fn dispatch<M: Matrix>(m: &M, mode: &str, c: &Case) -> String {
    let has = |k: &str| mode.split('+').any(|x| x == k);
    let (t, x, l, s, e) = (has("trace"), has("x_drop"), has("local_start"), has("free_query_start_gaps"), has("free_query_end_gaps"));
    macro_rules! go { ($t:literal, $x:literal, $l:literal, $s:literal, $e:literal) => { if (t, x, l, s, e) == ($t, $x, $l, $s, $e) { return run::<M, $t, $x, $l, $s, $e>(m, c); } } }
    macro_rules! both { ($x:literal, $l:literal, $s:literal, $e:literal) => { go!(false, $x, $l, $s, $e); go!(true, $x, $l, $s, $e); } }
    both!(false, false, false, false); both!(true, false, false, false);
    both!(false, true, false, false); both!(true, true, false, false);
    both!(false, false, true, false); both!(true, false, true, false);
    both!(false, false, false, true); both!(false, true, false, true); both!(false, false, true, true);
    panic!("mode {:?} is not one Block::align accepts", mode);
}

fn aa_static(name: &str) -> Option<&'static AAMatrix> {
    Some(match name {
        "BLOSUM45" => &BLOSUM45, "BLOSUM50" => &BLOSUM50, "BLOSUM62" => &BLOSUM62, "BLOSUM80" => &BLOSUM80, "BLOSUM90" => &BLOSUM90,
        "PAM100" => &PAM100, "PAM120" => &PAM120, "PAM160" => &PAM160, "PAM200" => &PAM200, "PAM250" => &PAM250,
        _ => return None
    })
}

fn simple(name: &str) -> Option<(i8, i8)> {
    let mut it = name.split(':');
    if it.next()? != "simple" { return None; }
    Some((it.next()?.parse().ok()?, it.next()?.parse().ok()?))
}

fn main() {
    let args: Vec<String> = env::args().collect();
    assert!(args.len() == 3, "usage: dump_golden <crate_golden_input.tsv> <crate_golden.tsv>");
    let text = fs::read_to_string(&args[1]).expect("input file");
    let mut out = fs::File::create(&args[2]).expect("output file");
    for line in text.lines() {
        if line.is_empty() || line.starts_with('#') { continue; }
        let f: Vec<&str> = line.split('\t').collect();
        assert!(f.len() == 11, "11 tab-separated fields per line: {:?}", line);
        let seq = |s: &'static str| -> &'static [u8] { if s == "-" { b"" } else { s.as_bytes() } };
        // (the fields borrow from `text`, which lives to the end of main: extend the borrow for the helper above)
        let (qs, rs): (&'static str, &'static str) = unsafe { (std::mem::transmute(f[9]), std::mem::transmute(f[10])) };
        let mode = if f[8] == "-" { "" } else { f[8] };
        if f[1] == "profile" {
            let max: usize = f[6].parse().unwrap();
            let c = Case { gaps: Gaps { open: -2, extend: -1 }, min: f[5].parse().unwrap(), max, x_drop: f[7].parse().unwrap(), q: seq(qs), r: seq(rs) };   // (gaps unused)
            let p = build_profile(seq(rs), f[3], f[4].parse().unwrap(), max);
            writeln!(out, "{}\t{}", f[0], dispatch_profile(&p, mode, &c)).unwrap();
            continue;
        }
        let c = Case { gaps: Gaps { open: f[3].parse().unwrap(), extend: f[4].parse().unwrap() }, min: f[5].parse().unwrap(), max: f[6].parse().unwrap(),
                       x_drop: f[7].parse().unwrap(), q: seq(qs), r: seq(rs) };
        let res = match f[1] {
            "aa" => match (aa_static(f[2]), simple(f[2])) {
                (Some(m), _) => dispatch(m, mode, &c),
                (None, Some((a, b))) => dispatch(&AAMatrix::new_simple(a, b), mode, &c),
                _ => panic!("unknown amino-acid matrix {:?}", f[2])
            },
            "nuc" => match (f[2], simple(f[2])) {
                ("NW1", _) => dispatch(&NW1, mode, &c),
                (_, Some((a, b))) => dispatch(&NucMatrix::new_simple(a, b), mode, &c),
                _ => panic!("unknown nucleotide matrix {:?}", f[2])
            },
            "bytes" => match (f[2], simple(f[2])) {
                ("BYTES1", _) => dispatch(&BYTES1, mode, &c),
                (_, Some((a, b))) => dispatch(&ByteMatrix::new_simple(a, b), mode, &c),
                _ => panic!("unknown byte matrix {:?}", f[2])
            },
            k => panic!("unknown kind {:?}", k)
        };
        writeln!(out, "{}\t{}", f[0], res).unwrap();
    }
}
