// make_reference_vectors.rs -- dumps golden .divans containers from the REAL reference build, so that one
// `cargo test` on any machine with a Rust toolchain pins the compressed bytes this repository can only pin by
// construction (no rustc in the build image; the reference holds no golden compressed vector of its own).
//
// How to run (from a checkout of dropbox/divans, the tree this repo calls /root/reference):
//   1. append this file to src/bin/benchmark.rs          (it reuses that module's TestSelection variants, its
//      cat make_reference_vectors.rs >> src/bin/benchmark.rs   ItemVecAllocator, init_shuffle_384 and recode_cmd_buffer)
//   2. cargo test --release --bin divans dump_reference_vectors -- --nocapture
//   3. copy the ref_container_*.divans files it writes (crate root) into tests/golden/ of THIS repository
// tests/test_reference_vectors.py then compares, byte for byte: the whole container against the oracle's, the LIT-coder
// stream inside it against the oracle's literal coder, and (on a GPU box) against the HIP kernels' output.
//
// Each container is what bench_no_ir (src/bin/benchmark.rs:292-343) compresses: the command list
// [PredictionMode, BlockSwitchLiteral(1, 2), Literal(data)] through DivansCompressorFactoryStruct with that variant's
// options and a 65 536-byte output buffer.  data = the shuffled-384 pattern (init_shuffle_384) repeated to `size` bytes.
// File name: ref_container_<variant>_<size>.divans
#[cfg(test)]
fn dump_one_reference_vector<TS: TestSelection>(ts: TS, name: &str) {
    use std::io::Write;
    let buffer_size = 65_536usize;
    let mut m8 = ItemVecAllocator::<u8>::default();
    let mut input_buffer = m8.alloc_cell(ts.size());
    let mut cmd_data_buffer = m8.alloc_cell(ts.size());
    let mut temp_buffer = m8.alloc_cell(buffer_size);
    let mut cm = m8.alloc_cell(256);
    let mut dm = m8.alloc_cell(PredictionModeContextMap::<ItemVec<u8>>::size_of_combined_array(256));
    for (index, item) in cm.slice_mut().iter_mut().enumerate() {
        *item = (index & 63) as u8;
    }
    let offset = PredictionModeContextMap::<ItemVec<u8>>::size_of_combined_array(0);
    for (index, item) in dm.slice_mut().iter_mut().enumerate() {
        if index >= offset {
            *item = ((index - offset) & 63) as u8;
        }
    }
    init_shuffle_384(input_buffer.slice_mut());
    cmd_data_buffer.slice_mut().clone_from_slice(input_buffer.slice());
    let mut pred_mode = PredictionModeContextMap {
        literal_context_map: cm,
        predmode_speed_and_distance_context_map: dm,
    };
    pred_mode.set_literal_prediction_mode(ts.prediction_mode());
    for item in pred_mode.get_mixing_values_mut().iter_mut() {
        *item = 4;
    }
    let ibuffer: [Command<ItemVec<u8>>; 3] = [
        Command::PredictionMode(pred_mode),
        Command::BlockSwitchLiteral(LiteralBlockSwitch::new(1, 2)),
        Command::Literal(LiteralCommand {
            data: cmd_data_buffer,
            prob: FeatureFlagSliceType::<ItemVec<u8>>::default(),
            high_entropy: false,
        }),
    ];
    let mut opts = divans::DivansCompressorOptions::default();
    opts.dynamic_context_mixing = Some(ts.adaptive_context_mixing() as u8 * 2);
    opts.prior_depth = ts.prior_depth();
    opts.use_context_map = ts.use_context_map();
    opts.force_stride_value = ts.stride_selection();
    opts.literal_adaptation = None;
    opts.window_size = Some(22);
    let mut encode_state = DivansCompressorFactoryStruct::<ItemVecAllocator<u8>, ItemVecAllocator<divans::DefaultCDF16>>::new(
        ItemVecAllocator::<u8>::default(),
        ItemVecAllocator::<u32>::default(),
        ItemVecAllocator::<divans::DefaultCDF16>::default(),
        opts,
        (),
    );
    let mut out = std::vec::Vec::<u8>::new();
    super::recode_cmd_buffer(&mut encode_state, &ibuffer[..], &mut out, temp_buffer.slice_mut()).unwrap();
    loop {
        let mut o_processed_index = 0;
        match encode_state.flush(temp_buffer.slice_mut(), &mut o_processed_index) {
            DivansOutputResult::Success => {
                out.extend_from_slice(temp_buffer.slice().split_at(o_processed_index).0);
                break;
            }
            DivansOutputResult::NeedsMoreOutput => {
                assert!(o_processed_index != 0);
                out.extend_from_slice(temp_buffer.slice().split_at(o_processed_index).0);
            }
            _ => panic!("Failure"),
        }
    }
    let path = format!("ref_container_{}_{}.divans", name, ts.size());
    std::fs::File::create(&path).unwrap().write_all(&out[..]).unwrap();
    println!("{}: {} -> {} bytes", path, ts.size(), out.len());
}

#[test]
fn dump_reference_vectors() {
    for size in [4097usize, 104_857usize].iter() {
        dump_one_reference_vector(TestSimple { size: *size }, "TestSimple");
        dump_one_reference_vector(TestAdapt { size: *size }, "TestAdapt");
        dump_one_reference_vector(TestContextMixing { size: *size }, "TestContextMixing");
        dump_one_reference_vector(TestContextMixingPureAverage { size: *size }, "TestContextMixingPureAverage");
    }
}
