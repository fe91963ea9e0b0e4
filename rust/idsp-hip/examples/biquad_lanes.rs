//! The reference's bit-exact i32 lowpass doctest (src/iir/coefficients.rs:289-300) replayed on 65536 GPU lanes
//! through the `dsp_process` view traits: every lane must print the reference's `[5, 3, 9, 25, 42, 49]`.
use dsp_fixedpoint::Q32;
use dsp_process::{LaneMajor, Split, ViewInplace, ViewProcess};
use idsp::iir::{Biquad, DirectForm1, coefficients::Filter};
use idsp_hip::{DevBuf, DevView, DevViewMut, GpuLanes, GpuState};

fn main() -> Result<(), Box<dyn std::error::Error>> {
    const LANES: usize = 65536;
    const FRAMES: usize = 6;
    // reference: let iir: Biquad<Q32<30>> = Filter::default().critical_frequency(0.1).gain(1000.0).lowpass().into();
    let biquad: Biquad<Q32<30>> = Filter::default().critical_frequency(0.1).gain(1000.0).lowpass().into();
    let lane = [3, -4, 5, 7, -3, 2];
    let host: Vec<i32> = (0..LANES).flat_map(|_| lane).collect(); // LaneMajor: each lane a contiguous slice
    let x = DevBuf::from_host(&host)?;
    let mut y = DevBuf::<i32>::zeroed(LANES * FRAMES)?;

    // reference: Split::new(biquad, DirectForm1::default()).lanes::<N>()
    let mut p = Split::new(GpuLanes::new(biquad), GpuState::<DirectForm1<i32>>::new(LANES)?);
    p.process_view(DevView::<_, LaneMajor>::from_flat(&x, LANES, FRAMES), DevViewMut::from_flat(&mut y, LANES, FRAMES));

    let mut out = vec![0i32; LANES * FRAMES];
    y.to_host(&mut out)?;
    assert!(out.chunks(FRAMES).all(|l| l == [5, 3, 9, 25, 42, 49]));
    println!("lane 0: {:?}", &out[..FRAMES]);

    // the states came back like `[DirectForm1<i32>; N]` would: x = [x0, x1], y = [[y0, y1]]
    let st = p.state.download()?;
    assert_eq!((st[0].x, st[0].y), ([2, -3], [[49, 42]]));

    // `inplace_view` continues the stream from that state
    let mut xy = DevBuf::from_host(&host)?;
    p.inplace_view(DevViewMut::<_, LaneMajor>::from_flat(&mut xy, LANES, FRAMES));
    Ok(())
}
