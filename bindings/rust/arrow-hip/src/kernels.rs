//! `arrow_hip::kernels` — the DROP-IN layer: the reference's own function signatures over arrays whose buffers live in
//! MI355X HBM.
//!
//! NOT COMPILED in the image this repository is developed in (no Rust toolchain there).  INTEGRATION.md §2 quotes the
//! blocks between the `integration:` markers of this file VERBATIM; `tests/test_abi_symbols.py::
//! test_integration_md_quotes_the_crate` fails when the document and the crate drift apart, and
//! `test_rust_sys_bindings_are_current` checks every `sys::` name used here against the generated declarations.
//!
//! | here | reference signature |
//! |---|---|
//! | [`filter`] | `arrow_select::filter::filter(&dyn Array, &BooleanArray) -> Result<ArrayRef, ArrowError>` (filter.rs:201) |
//! | [`take`] | `arrow_select::take::take(&dyn Array, &dyn Array, Option<TakeOptions>) -> Result<ArrayRef, ArrowError>` (take.rs:89) |
//! | [`add_wrapping`] … [`rem`] | `arrow_arith::numeric::*(&dyn Datum, &dyn Datum) -> Result<ArrayRef, ArrowError>` (numeric.rs:36-81) |
//! | [`lt`] … [`not_distinct`] | `arrow_ord::cmp::*(&dyn Datum, &dyn Datum) -> Result<BooleanArray, ArrowError>` (cmp.rs:79-202) |
//! | [`cast`] | `arrow_cast::cast(&dyn Array, &DataType) -> Result<ArrayRef, ArrowError>` (cast/mod.rs:347) |
//!
//! The contract that makes the signatures work: an `ArrayRef` handled by this module is an ordinary arrow-rs array
//! whose `Buffer`s were made with `Buffer::from_custom_allocation` (arrow-buffer/src/buffer/immutable.rs:170) over
//! DEVICE pointers.  Nothing here ever dereferences one on the host: results are assembled with the constructors that
//! only record pointers and lengths — `NullBuffer::new_unchecked` (buffer/null.rs:70), `BooleanBuffer::new`
//! (buffer/boolean.rs:124), `ScalarBuffer::new` (buffer/scalar.rs:89), `ArrayDataBuilder::build_unchecked`
//! (arrow-data/src/data.rs:2213) — and `null_count` comes from the kernel, not from counting bits.  Host code that
//! READS such an array (`PartialEq`, `Debug`, iterators) must first bring it back with [`download`].  Scalar operands
//! (`Datum::get().1 == true`) are the exception: a `Scalar` is one value the caller built on the host
//! (`Int64Array::new_scalar(5)`), so its bytes are read here and travel in a cached device word.
use std::cell::RefCell;
use std::ffi::c_void;
use std::mem::MaybeUninit;
use std::ptr::{self, NonNull};
use std::sync::Arc;

use arrow::array::{make_array, Array, ArrayData, ArrayRef, BooleanArray, Datum};
use arrow::buffer::{BooleanBuffer, Buffer, NullBuffer};
use arrow::datatypes::DataType;
use arrow::error::ArrowError;
use arrow_hip_sys as sys;

use crate::{Context, DeviceArray};

pub use arrow::compute::TakeOptions;

thread_local! {
    // one context (= one HIP stream + pooled allocator) per calling thread, on the device `ARROW_HIP_DEVICE` names
    static CURRENT: RefCell<Option<Arc<Context>>> = const { RefCell::new(None) };
}

/// The calling thread's context (created on first use).
pub fn current_context() -> Result<Arc<Context>, ArrowError> {
    CURRENT.with(|c| {
        if let Some(ctx) = c.borrow().as_ref() {
            return Ok(ctx.clone());
        }
        let device = std::env::var("ARROW_HIP_DEVICE").ok().and_then(|d| d.parse().ok()).unwrap_or(0);
        let ctx = Context::new(device)?;
        *c.borrow_mut() = Some(ctx.clone());
        Ok(ctx)
    })
}

// <<integration:owner
/// Owner of one kernel result: every `Buffer` cut from it holds an `Arc` of this, and the last one to go returns the
/// HBM to the library (`ah_array_release`).  `Allocation` is blanket-implemented for `RefUnwindSafe + Send + Sync`
/// types (arrow-buffer/src/alloc/mod.rs:31-33).
struct HbmAllocation {
    ctx: Arc<Context>,
    out: std::sync::Mutex<sys::ah_array_out>,
    /// inputs a zero-copy result still points into (AH_OUT_BORROWED: filter's `All` strategy returns `values.slice`)
    _keep: Vec<ArrayRef>,
}
unsafe impl Send for HbmAllocation {}
unsafe impl Sync for HbmAllocation {}
impl std::panic::RefUnwindSafe for HbmAllocation {}
impl Drop for HbmAllocation {
    fn drop(&mut self) {
        unsafe { sys::ah_array_release(self.ctx.raw(), &mut *self.out.lock().unwrap()) }
    }
}
// integration>>

fn ah_type_of(dt: &DataType) -> Result<i32, ArrowError> {
    use DataType::*;
    Ok(match dt {
        Boolean => sys::AH_BOOL,
        Int8 => sys::AH_INT8,
        Int16 => sys::AH_INT16,
        Int32 | Date32 | Time32(_) => sys::AH_INT32,
        Int64 | Date64 | Time64(_) | Timestamp(_, _) | Duration(_) => sys::AH_INT64,
        UInt8 => sys::AH_UINT8,
        UInt16 => sys::AH_UINT16,
        UInt32 => sys::AH_UINT32,
        UInt64 => sys::AH_UINT64,
        Float16 => sys::AH_FLOAT16,
        Float32 => sys::AH_FLOAT32,
        Float64 => sys::AH_FLOAT64,
        Decimal128(_, _) => sys::AH_FIXED16,
        Decimal256(_, _) => sys::AH_FIXED32,
        Utf8 => sys::AH_UTF8,
        LargeUtf8 => sys::AH_LARGE_UTF8,
        other => return Err(ArrowError::NotYetImplemented(format!("arrow-hip kernels for {other}"))),
    })
}

// <<integration:view
/// `&dyn Array` -> the C ABI's borrowed view.  Only pointers, offsets and counts are taken: no byte is read.
fn view_of(array: &dyn Array) -> Result<sys::ah_array_view, ArrowError> {
    let data: ArrayData = array.to_data();
    let t = ah_type_of(data.data_type())?;
    let (off, bufs) = (data.offset(), data.buffers());
    let mut v = sys::ah_array_view {
        type_: t,
        length: data.len() as i64,
        null_count: data.nulls().map_or(0, |n| n.null_count() as i64),
        values: ptr::null(),
        values_bit_offset: 0,
        validity: data.nulls().map_or(ptr::null(), |n| n.validity().as_ptr()),
        validity_bit_offset: data.nulls().map_or(0, |n| n.offset() as i64),
        offsets: ptr::null(),
    };
    match data.data_type() {
        DataType::Boolean => {
            v.values = bufs[0].as_ptr() as *const c_void;
            v.values_bit_offset = off as i64;
        }
        DataType::Utf8 | DataType::LargeUtf8 => {
            let w = if t == sys::AH_UTF8 { 4 } else { 8 };
            v.offsets = unsafe { bufs[0].as_ptr().add(off * w) } as *const c_void; // the offsets pointer is advanced,
            v.values = bufs[1].as_ptr() as *const c_void; //                          the data pointer is not (array slice semantics)
        }
        dt => {
            let w = dt.primitive_width().expect("fixed-width type");
            v.values = unsafe { bufs[0].as_ptr().add(off * w) } as *const c_void;
        }
    }
    Ok(v)
}
// integration>>

// <<integration:wrap
/// `ah_array_out` -> `ArrayRef` without touching a byte: buffers are `Buffer::from_custom_allocation` over the device
/// pointers, all sharing one [`HbmAllocation`]; `null_count` is the kernel's.
fn wrap(ctx: Arc<Context>, out: sys::ah_array_out, data_type: DataType, inputs: &[&dyn Array]) -> Result<ArrayRef, ArrowError> {
    let len = out.length as usize;
    let borrowed = out.flags & (sys::AH_OUT_BORROWED | sys::AH_OUT_BORROWED_VALUES) != 0;
    let keep = if borrowed { inputs.iter().map(|a| make_array(a.to_data())).collect() } else { vec![] };
    let owner: Arc<HbmAllocation> = Arc::new(HbmAllocation { ctx, out: std::sync::Mutex::new(out), _keep: keep });
    let buffer = |p: *mut c_void, bytes: i64| -> Buffer {
        match NonNull::new(p as *mut u8) {
            // SAFETY: the library guarantees `bytes` bytes of HBM at `p` until ah_array_release, which only `owner` calls
            Some(nn) => unsafe { Buffer::from_custom_allocation(nn, bytes as usize, owner.clone()) },
            None => Buffer::from(Vec::<u8>::new()), // an empty result carries no allocation
        }
    };
    let nulls = (!out.validity.is_null()).then(|| {
        let bits = BooleanBuffer::new(buffer(out.validity as *mut c_void, out.validity_bytes), out.validity_bit_offset as usize, len);
        // SAFETY: `null_count` was counted by the kernel that wrote these bits
        unsafe { NullBuffer::new_unchecked(bits, out.null_count as usize) }
    });
    let mut b = ArrayData::builder(data_type.clone()).len(len).nulls(nulls);
    b = match &data_type {
        DataType::Boolean => b.offset(out.values_bit_offset as usize).add_buffer(buffer(out.values, out.values_bytes)),
        DataType::Utf8 | DataType::LargeUtf8 => b.add_buffer(buffer(out.offsets, out.offsets_bytes)).add_buffer(buffer(out.values, out.values_bytes)),
        _ => b.add_buffer(buffer(out.values, out.values_bytes)), // `values_bytes` may exceed len * width (worst-case allocation)
    };
    // SAFETY: layout as documented in include/arrow_hip.h; validation would READ offsets / bits on the host
    Ok(make_array(unsafe { b.build_unchecked() }))
}
// integration>>

fn call(ctx: &Arc<Context>, f: impl FnOnce(*mut sys::ah_context, *mut sys::ah_array_out) -> sys::ah_status) -> Result<sys::ah_array_out, ArrowError> {
    let mut out = MaybeUninit::<sys::ah_array_out>::zeroed();
    ctx.check(f(ctx.raw(), out.as_mut_ptr()))?; // status -> the reference's ArrowError variant and text; its panics stay panics
    Ok(unsafe { out.assume_init() })
}

// <<integration:filter
/// `arrow_select::filter::filter` (arrow-select/src/filter.rs:201), same signature.
pub fn filter(values: &dyn Array, predicate: &BooleanArray) -> Result<ArrayRef, ArrowError> {
    let ctx = current_context()?;
    let (v, p) = (view_of(values)?, view_of(predicate)?);
    let out = call(&ctx, |c, o| unsafe { sys::ah_filter(c, &v, &p, o) })?;
    wrap(ctx, out, values.data_type().clone(), &[values])
}
// integration>>

// <<integration:take
/// `arrow_select::take::take` (arrow-select/src/take.rs:89), same signature; `TakeOptions` is the reference's own type.
pub fn take(values: &dyn Array, indices: &dyn Array, options: Option<TakeOptions>) -> Result<ArrayRef, ArrowError> {
    let ctx = current_context()?;
    let (v, i) = (view_of(values)?, view_of(indices)?);
    let check = options.unwrap_or_default().check_bounds as i32;
    let out = call(&ctx, |c, o| unsafe { sys::ah_take(c, &v, &i, check, o) })?;
    wrap(ctx, out, values.data_type().clone(), &[])
}
// integration>>

/// One operand of a binary kernel: `Datum::get()` (arrow-array/src/scalar.rs:78-98).  An array operand is device
/// resident; a scalar operand is the caller's one-element HOST array, staged into a device word for the call.
struct Operand {
    view: sys::ah_array_view,
    is_scalar: i32,
    _staged: Option<Arc<DeviceArray>>,
}

fn operand(ctx: &Arc<Context>, d: &dyn Datum) -> Result<Operand, ArrowError> {
    let (array, is_scalar) = d.get();
    if !is_scalar {
        return Ok(Operand { view: view_of(array)?, is_scalar: 0, _staged: None });
    }
    let staged = DeviceArray::from_host(ctx, array)?; // one value: `to_ffi` + ah_import_c_data
    Ok(Operand { view: staged.view(), is_scalar: 1, _staged: Some(staged) })
}

// <<integration:numeric
fn arithmetic_op(op: i32, lhs: &dyn Datum, rhs: &dyn Datum) -> Result<ArrayRef, ArrowError> {
    let ctx = current_context()?;
    let (l, r) = (operand(&ctx, lhs)?, operand(&ctx, rhs)?);
    let out = call(&ctx, |c, o| unsafe { sys::ah_arith_binary(c, op, &l.view, l.is_scalar, &r.view, r.is_scalar, o) })?;
    wrap(ctx, out, lhs.get().0.data_type().clone(), &[])
}

/// `arrow_arith::numeric::add_wrapping` (arrow-arith/src/numeric.rs:41), same signature.
pub fn add_wrapping(lhs: &dyn Datum, rhs: &dyn Datum) -> Result<ArrayRef, ArrowError> {
    arithmetic_op(sys::AH_ADD_WRAPPING, lhs, rhs)
}
// integration>>

macro_rules! numeric_fn {
    ($($(#[$doc:meta])* $name:ident => $op:ident),* $(,)?) => {$(
        $(#[$doc])*
        pub fn $name(lhs: &dyn Datum, rhs: &dyn Datum) -> Result<ArrayRef, ArrowError> { arithmetic_op(sys::$op, lhs, rhs) }
    )*};
}
numeric_fn! {
    /// `arrow_arith::numeric::add` (numeric.rs:36): checked for integers (`ArithmeticOverflow` with the reference's text)
    add => AH_ADD,
    /// `sub` (numeric.rs:46)
    sub => AH_SUB,
    /// `sub_wrapping` (numeric.rs:51)
    sub_wrapping => AH_SUB_WRAPPING,
    /// `mul` (numeric.rs:56)
    mul => AH_MUL,
    /// `mul_wrapping` (numeric.rs:61)
    mul_wrapping => AH_MUL_WRAPPING,
    /// `div` (numeric.rs:69): `DivideByZero` for integers, IEEE for floats
    div => AH_DIV,
    /// `rem` (numeric.rs:79)
    rem => AH_REM,
}

// <<integration:cmp
fn compare_op(op: i32, lhs: &dyn Datum, rhs: &dyn Datum) -> Result<BooleanArray, ArrowError> {
    let ctx = current_context()?;
    let (l, r) = (operand(&ctx, lhs)?, operand(&ctx, rhs)?);
    // compare_op's rule that the LOGICAL types agree (cmp.rs:243-264: Decimal128(12, 3) vs Decimal128(12, 1) is refused) is the
    // library's (ah_compare_with_types): this layer only says which logical types it holds
    let (lt, rt) = (crate::logical(lhs.get().0.data_type())?, crate::logical(rhs.get().0.data_type())?);
    let out = if lt.is_none() && rt.is_none() {
        call(&ctx, |c, o| unsafe { sys::ah_compare(c, op, &l.view, l.is_scalar, &r.view, r.is_scalar, o) })?
    } else {
        let plain = |id| sys::ah_data_type { id, unit: 0, has_tz: 0, tz_offset_seconds: 0, precision: 0, scale: 0 };
        let (lt, rt) = (lt.unwrap_or(plain(l.view.type_)), rt.unwrap_or(plain(r.view.type_)));
        call(&ctx, |c, o| unsafe { sys::ah_compare_with_types(c, op, &l.view, l.is_scalar, &lt, &r.view, r.is_scalar, &rt, o) })?
    };
    Ok(BooleanArray::from(wrap(ctx, out, DataType::Boolean, &[])?.to_data()))
}

/// `arrow_ord::cmp::lt` (arrow-ord/src/cmp.rs:113), same signature: totalOrder for floats.
pub fn lt(lhs: &dyn Datum, rhs: &dyn Datum) -> Result<BooleanArray, ArrowError> {
    compare_op(sys::AH_LT, lhs, rhs)
}
// integration>>

macro_rules! cmp_fn {
    ($($(#[$doc:meta])* $name:ident => $op:ident),* $(,)?) => {$(
        $(#[$doc])*
        pub fn $name(lhs: &dyn Datum, rhs: &dyn Datum) -> Result<BooleanArray, ArrowError> { compare_op(sys::$op, lhs, rhs) }
    )*};
}
cmp_fn! {
    /// `eq` (cmp.rs:79): bit equality for floats
    eq => AH_EQ,
    /// `neq` (cmp.rs:96)
    neq => AH_NEQ,
    /// `lt_eq` (cmp.rs:130)
    lt_eq => AH_LT_EQ,
    /// `gt` (cmp.rs:147)
    gt => AH_GT,
    /// `gt_eq` (cmp.rs:164)
    gt_eq => AH_GT_EQ,
    /// `distinct` (cmp.rs:182): never null
    distinct => AH_DISTINCT,
    /// `not_distinct` (cmp.rs:200)
    not_distinct => AH_NOT_DISTINCT,
}

// <<integration:cast
/// `arrow_cast::cast` (arrow-cast/src/cast/mod.rs:347), same signature (`CastOptions::default()`: safe).
pub fn cast(array: &dyn Array, to_type: &DataType) -> Result<ArrayRef, ArrowError> {
    let ctx = current_context()?;
    let v = view_of(array)?;
    let to = ah_type_of(to_type)?;
    let out = call(&ctx, |c, o| unsafe { sys::ah_cast(c, &v, to, 1, o) })?;
    wrap(ctx, out, to_type.clone(), &[])
}
// integration>>

/// Host array -> an `ArrayRef` of the same type whose buffers live in HBM (the explicit upload; `to_ffi` +
/// `ah_import_c_data`).
pub fn upload(array: &dyn Array) -> Result<ArrayRef, ArrowError> {
    let ctx = current_context()?;
    let out = DeviceArray::import(&ctx, array)?; // the owned ah_array_out: `wrap` hands it to an HbmAllocation
    wrap(ctx, out, array.data_type().clone(), &[])
}

/// A device-resident `ArrayRef` -> an ordinary host array (`ah_export_c_data` + `from_ffi`): the explicit download.
pub fn download(array: &dyn Array) -> Result<ArrayRef, ArrowError> {
    let ctx = current_context()?;
    DeviceArray::borrowing(&ctx, view_of(array)?, array.data_type().clone()).to_host()
}
