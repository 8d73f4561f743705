//! arrow-hip — `arrow::compute::kernels`-shaped functions whose arrays live in MI355X HBM.
//!
//! NOT COMPILED in the image this repository is developed in (no Rust toolchain there); see ../README.md.
//! Every function is a thin call into `libarrow_hip.so` (declarations: `arrow-hip-sys`, generated from
//! `include/arrow_hip.h`).  Names, argument order and error behaviour follow the reference:
//!
//! | here | reference |
//! |---|---|
//! | [`filter`], [`FilterPredicate`] | `arrow_select::filter::{filter, FilterBuilder, FilterPredicate}` (arrow-select/src/filter.rs:201,:248) |
//! | [`take`] | `arrow_select::take::take` (arrow-select/src/take.rs:89) |
//! | [`add`], [`add_wrapping`], … [`rem`], [`neg`] | `arrow_arith::numeric` (arrow-arith/src/numeric.rs:36-186) |
//! | [`eq`] … [`not_distinct`] | `arrow_ord::cmp` (arrow-ord/src/cmp.rs:79-202) |
//! | [`cast`], [`cast_with_options`] | `arrow_cast::cast` (arrow-cast/src/cast/mod.rs:347,:790) |
//! | [`and`], [`or`], [`not`], [`is_null`], … | `arrow_arith::boolean` (arrow-arith/src/boolean.rs:60-360) |
//! | [`concat`] | `arrow_select::concat::concat` (arrow-select/src/concat.rs:495) |
//! | [`shift`] | `arrow_select::window::shift` (arrow-select/src/window.rs:56) |
//! | [`rank`] | `arrow_ord::rank::rank` (arrow-ord/src/rank.rs:58) |
//! | [`DeviceArray::from_host`], [`DeviceArray::to_host`] | `arrow::ffi::{to_ffi, from_ffi}` (arrow-array/src/ffi.rs:231-254) |
//! | [`BatchCoalescer`] | `arrow_select::coalesce::BatchCoalescer` (arrow-select/src/coalesce.rs:148) |
//! | [`Comm`] | — (no parallelism in the reference): `concat` / `concat_batches` (concat.rs:495,:607) of row shards across GPUs |
use std::ffi::CStr;
use std::mem::MaybeUninit;
use std::ptr;
use std::sync::Arc;

use arrow::array::{make_array, Array, ArrayData, ArrayRef};
use arrow::compute::SortOptions;
use arrow::datatypes::DataType;
use arrow::error::ArrowError;
use arrow::ffi::{from_ffi, to_ffi, FFI_ArrowArray, FFI_ArrowSchema};
use arrow_hip_sys as sys;

/// The drop-in layer: the reference's own signatures (`filter(&dyn Array, &BooleanArray) -> Result<ArrayRef>`, …) over
/// arrays whose buffers are `Buffer::from_custom_allocation` views of HBM.
pub mod kernels;

/// One HIP stream + pooled HBM allocator on one GPU.
pub struct Context {
    raw: *mut sys::ah_context,
}

// every entry point of libarrow_hip.so locks its context for the duration of the call (a recursive mutex inside
// `ah_context`), so one `Context` may be shared by threads: calls on it serialise.  Threads that
// want their kernels to overlap create one context each.
unsafe impl Send for Context {}
unsafe impl Sync for Context {}

impl Context {
    pub fn new(device: i32) -> Result<Arc<Self>, ArrowError> {
        let mut raw = ptr::null_mut();
        let st = unsafe { sys::ah_context_create(device, &mut raw) };
        if st != sys::AH_OK {
            return Err(ArrowError::ExternalError(
                format!("ah_context_create(device={device}) failed with status {st}: no usable MI355X").into(),
            ));
        }
        Ok(Arc::new(Self { raw }))
    }

    /// Opt-in asynchronous calls (`ah_context_set_deferred`, include/arrow_hip.h).
    pub fn set_deferred(&self, on: bool) {
        unsafe { sys::ah_context_set_deferred(self.raw, on as i32) }
    }

    pub fn synchronize(&self) -> Result<(), ArrowError> {
        self.check(unsafe { sys::ah_synchronize(self.raw) })
    }

    pub(crate) fn raw(&self) -> *mut sys::ah_context {
        self.raw
    }

    fn message(&self) -> String {
        unsafe { CStr::from_ptr(sys::ah_last_error(self.raw)) }.to_string_lossy().into_owned()
    }

    /// status -> `ArrowError` with the reference's message text; reference panics stay panics.
    pub(crate) fn check(&self, st: sys::ah_status) -> Result<(), ArrowError> {
        match st {
            sys::AH_OK => Ok(()),
            sys::AH_INVALID_ARGUMENT => Err(ArrowError::InvalidArgumentError(self.message())),
            sys::AH_COMPUTE_ERROR => Err(ArrowError::ComputeError(self.message())),
            sys::AH_ARITHMETIC_OVERFLOW => Err(ArrowError::ArithmeticOverflow(self.message())),
            sys::AH_DIVIDE_BY_ZERO => Err(ArrowError::DivideByZero),
            sys::AH_CAST_ERROR => Err(ArrowError::CastError(self.message())),
            sys::AH_NOT_YET_IMPLEMENTED => Err(ArrowError::NotYetImplemented(self.message())),
            sys::AH_OFFSET_OVERFLOW_ERROR => {
                Err(ArrowError::OffsetOverflowError(self.message().parse().unwrap_or(usize::MAX)))
            }
            sys::AH_C_DATA_INTERFACE => Err(ArrowError::CDataInterface(self.message())),
            sys::AH_IPC_ERROR => Err(ArrowError::IpcError(self.message())),
            sys::AH_PARSE_ERROR => Err(ArrowError::ParseError(self.message())),
            sys::AH_OFFSET_OVERFLOW | sys::AH_PANIC => panic!("{}", self.message()),
            _ => Err(ArrowError::ExternalError(self.message().into())),
        }
    }
}

impl Drop for Context {
    fn drop(&mut self) {
        unsafe { sys::ah_context_destroy(self.raw) }
    }
}

/// A recorded sequence of deferred calls (`ah_graph_begin` / `_end`): `Context::capture(|| { … })` records the
/// kernels the closure enqueues instead of running them; [`Graph::launch`] replays them as ONE hipGraph launch over the
/// current bytes of the captured inputs, into the outputs the recorded calls returned (keep those alive).
pub struct Graph {
    ctx: Arc<Context>,
    raw: *mut sys::ah_graph,
}

impl Graph {
    pub fn launch(&self) -> Result<(), ArrowError> {
        self.ctx.check(unsafe { sys::ah_graph_launch(self.ctx.raw, self.raw) })
    }

    pub fn node_count(&self) -> usize {
        unsafe { sys::ah_graph_node_count(self.raw) as usize }
    }
}

impl Drop for Graph {
    fn drop(&mut self) {
        unsafe { sys::ah_graph_destroy(self.ctx.raw, self.raw) }
    }
}

impl Context {
    /// Record the deferred calls `record` makes on this context (wrapping / float arithmetic, cmp, boolean, safe numeric
    /// casts, `FilterPredicate::filter` with a prebuilt predicate); a call that must wait on the device fails fast.
    pub fn capture<R>(self: &Arc<Self>, record: impl FnOnce() -> Result<R, ArrowError>) -> Result<(Graph, R), ArrowError> {
        self.check(unsafe { sys::ah_graph_begin(self.raw) })?;
        let recorded = record();
        let mut raw = ptr::null_mut();
        let st = unsafe { sys::ah_graph_end(self.raw, &mut raw) };
        match recorded {
            Ok(r) => {
                self.check(st)?;
                Ok((Graph { ctx: self.clone(), raw }, r))
            }
            Err(e) => {
                if st == sys::AH_OK {
                    unsafe { sys::ah_graph_destroy(self.raw, raw) };
                }
                Err(e)
            }
        }
    }

    /// `MemoryPool::used` and friends for the pooled device allocator (`ah_context_stats`).
    pub fn memory_stats(&self, reset_peaks: bool) -> Result<sys::ah_context_stats_t, ArrowError> {
        let mut st = MaybeUninit::<sys::ah_context_stats_t>::zeroed();
        self.check(unsafe { sys::ah_context_stats(self.raw, st.as_mut_ptr(), reset_peaks as i32) })?;
        Ok(unsafe { st.assume_init() })
    }
}

/// An array whose buffers live in HBM.  Owns an `ah_array_out`; the logical `DataType` travels on the host
/// exactly as the reference carries `data_type` through filter / take (filter.rs:783-787, take.rs:414).
pub struct DeviceArray {
    ctx: Arc<Context>,
    out: sys::ah_array_out,
    data_type: DataType,
    /// inputs a zero-copy result still points into (AH_OUT_BORROWED / AH_OUT_BORROWED_VALUES)
    _keep: Vec<Arc<DeviceArray>>,
}

impl Drop for DeviceArray {
    fn drop(&mut self) {
        unsafe { sys::ah_array_release(self.ctx.raw, &mut self.out) }
    }
}

impl DeviceArray {
    pub fn len(&self) -> usize {
        self.out.length as usize
    }

    pub fn is_empty(&self) -> bool {
        self.out.length == 0
    }

    pub fn data_type(&self) -> &DataType {
        &self.data_type
    }

    /// `Array::null_count`; a deferred result (-1) is counted on first use.
    pub fn null_count(&mut self) -> Result<usize, ArrowError> {
        if self.out.null_count < 0 {
            self.ctx.check(unsafe { sys::ah_array_resolve(self.ctx.raw, &mut self.out) })?;
        }
        Ok(self.out.null_count as usize)
    }

    pub(crate) fn view(&self) -> sys::ah_array_view {
        sys::ah_array_view {
            type_: self.out.type_,
            length: self.out.length,
            null_count: if self.out.validity.is_null() { 0 } else { self.out.null_count },
            values: self.out.values,
            values_bit_offset: self.out.values_bit_offset,
            validity: self.out.validity,
            validity_bit_offset: self.out.validity_bit_offset,
            offsets: self.out.offsets,
        }
    }

    /// Host array -> HBM through the C Data Interface (`to_ffi`, then `ah_import_c_data` = `from_ffi` on the device).
    pub fn from_host(ctx: &Arc<Context>, array: &dyn Array) -> Result<Arc<Self>, ArrowError> {
        let out = Self::import(ctx, array)?;
        Ok(Arc::new(Self { ctx: ctx.clone(), out, data_type: array.data_type().clone(), _keep: vec![] }))
    }

    /// the raw half of `from_host`: the owned `ah_array_out` (whoever holds it calls `ah_array_release`)
    pub(crate) fn import(ctx: &Arc<Context>, array: &dyn Array) -> Result<sys::ah_array_out, ArrowError> {
        let (ffi_array, ffi_schema) = to_ffi(&array.to_data())?;
        let mut out = MaybeUninit::<sys::ah_array_out>::zeroed();
        ctx.check(unsafe {
            sys::ah_import_c_data(
                ctx.raw,
                &ffi_array as *const FFI_ArrowArray as *const sys::ArrowArray,
                &ffi_schema as *const FFI_ArrowSchema as *const sys::ArrowSchema,
                out.as_mut_ptr(),
            )
        })?;
        Ok(unsafe { out.assume_init() })
    }

    /// A `DeviceArray` that only BORROWS device buffers described by a view (`AH_OUT_BORROWED`: its release frees
    /// nothing) — how `kernels::download` exports an `ArrayRef` whose buffers are owned elsewhere.
    pub(crate) fn borrowing(ctx: &Arc<Context>, v: sys::ah_array_view, data_type: DataType) -> Self {
        let out = sys::ah_array_out {
            type_: v.type_,
            length: v.length,
            null_count: v.null_count,
            values: v.values as *mut _,
            values_bytes: 0,
            values_bit_offset: v.values_bit_offset,
            validity: v.validity as *mut _,
            validity_bytes: 0,
            validity_bit_offset: v.validity_bit_offset,
            offsets: v.offsets as *mut _,
            offsets_bytes: 0,
            flags: sys::AH_OUT_BORROWED,
        };
        Self { ctx: ctx.clone(), out, data_type, _keep: vec![] }
    }

    /// HBM -> host array (`ah_export_c_data` fills FFI structs whose release callbacks free the host copies).
    pub fn to_host(&self) -> Result<ArrayRef, ArrowError> {
        let schema = FFI_ArrowSchema::try_from(&self.data_type)?;
        let (mut out_array, mut out_schema) = (FFI_ArrowArray::empty(), FFI_ArrowSchema::empty());
        let view = self.view();
        self.ctx.check(unsafe {
            sys::ah_export_c_data(
                self.ctx.raw,
                &view,
                schema.format().as_ptr() as *const _,
                &mut out_array as *mut FFI_ArrowArray as *mut sys::ArrowArray,
                &mut out_schema as *mut FFI_ArrowSchema as *mut sys::ArrowSchema,
            )
        })?;
        let data: ArrayData = unsafe { from_ffi(out_array, &out_schema)? };
        Ok(make_array(data))
    }
}

fn wrap(like: &Arc<DeviceArray>, out: sys::ah_array_out, data_type: DataType, inputs: &[&Arc<DeviceArray>]) -> Arc<DeviceArray> {
    let borrowed = out.flags & (sys::AH_OUT_BORROWED | sys::AH_OUT_BORROWED_VALUES) != 0;
    let keep = if borrowed { inputs.iter().map(|a| (*a).clone()).collect() } else { vec![] };
    Arc::new(DeviceArray { ctx: like.ctx.clone(), out, data_type, _keep: keep })
}

/// `Datum::get()` (arrow-array/src/scalar.rs:78-98): an array, or a length-1 array standing for a scalar.
pub enum Datum<'a> {
    Array(&'a Arc<DeviceArray>),
    Scalar(&'a Arc<DeviceArray>),
}

impl<'a> Datum<'a> {
    fn get(&self) -> (&'a Arc<DeviceArray>, i32) {
        match self {
            Datum::Array(a) => (a, 0),
            Datum::Scalar(a) => (a, 1),
        }
    }
}

// ------------------------------------------------------------------------------------------- filter / take
/// `arrow_select::filter::filter` (filter.rs:201)
pub fn filter(values: &Arc<DeviceArray>, predicate: &Arc<DeviceArray>) -> Result<Arc<DeviceArray>, ArrowError> {
    let mut out = MaybeUninit::<sys::ah_array_out>::zeroed();
    let (v, p) = (values.view(), predicate.view());
    values.ctx.check(unsafe { sys::ah_filter(values.ctx.raw, &v, &p, out.as_mut_ptr()) })?;
    Ok(wrap(values, unsafe { out.assume_init() }, values.data_type.clone(), &[values]))
}

/// `FilterBuilder::new(p).optimize().build()` (filter.rs:256-324): count once, apply to many columns.
pub struct FilterPredicate {
    ctx: Arc<Context>,
    raw: *mut sys::ah_filter_predicate,
    _mask: Arc<DeviceArray>,
}

impl FilterPredicate {
    pub fn new(predicate: &Arc<DeviceArray>) -> Result<Self, ArrowError> {
        let mut raw = ptr::null_mut();
        let p = predicate.view();
        predicate.ctx.check(unsafe { sys::ah_filter_predicate_build(predicate.ctx.raw, &p, &mut raw) })?;
        Ok(Self { ctx: predicate.ctx.clone(), raw, _mask: predicate.clone() })
    }

    /// `FilterPredicate::count` (filter.rs:481)
    pub fn count(&self) -> usize {
        unsafe { sys::ah_filter_predicate_count(self.raw) as usize }
    }

    /// `FilterPredicate::filter` (filter.rs:473)
    pub fn filter(&self, values: &Arc<DeviceArray>) -> Result<Arc<DeviceArray>, ArrowError> {
        let mut out = MaybeUninit::<sys::ah_array_out>::zeroed();
        let v = values.view();
        self.ctx.check(unsafe { sys::ah_filter_predicate_apply(self.ctx.raw, self.raw, &v, out.as_mut_ptr()) })?;
        Ok(wrap(values, unsafe { out.assume_init() }, values.data_type.clone(), &[values]))
    }
}

impl Drop for FilterPredicate {
    fn drop(&mut self) {
        unsafe { sys::ah_filter_predicate_free(self.ctx.raw, self.raw) }
    }
}

/// One comparison of a filter expression: `op(lhs, rhs)` (`sys::AH_EQ` .. `sys::AH_GT_EQ`; cmp.rs:79-202).
pub struct Term<'a> {
    pub op: i32,
    pub lhs: Datum<'a>,
    pub rhs: Datum<'a>,
}

fn with_terms<R>(terms: &[Term], joins: &[i32],
                 f: impl FnOnce(&Arc<Context>, i32, *const sys::ah_filter_term, *const i32) -> Result<R, ArrowError>) -> Result<R, ArrowError> {
    if terms.is_empty() || joins.len() + 1 != terms.len() {
        return Err(ArrowError::InvalidArgumentError("a filter expression of n terms takes n - 1 joins".into()));
    }
    // the views must outlive the call: collect them first, then point the C terms at them
    let views: Vec<(sys::ah_array_view, i32, sys::ah_array_view, i32)> = terms.iter().map(|t| {
        let ((l, ls), (r, rs)) = (t.lhs.get(), t.rhs.get());
        (l.view(), ls, r.view(), rs)
    }).collect();
    let raw: Vec<sys::ah_filter_term> = terms.iter().zip(&views).map(|(t, v)| sys::ah_filter_term {
        op: t.op, lhs: &v.0, lhs_is_scalar: v.1, rhs: &v.2, rhs_is_scalar: v.3,
    }).collect();
    let ctx = &terms[0].lhs.get().0.ctx;
    f(ctx, raw.len() as i32, raw.as_ptr(), joins.as_ptr())
}

/// `filter(values, &and_kleene(&lt(a, x)?, &gt_eq(b, y)?)?)` with nothing materialised (cmp.rs:113-164,
/// arrow-arith/src/boolean.rs:60-300, filter.rs:201): the comparisons run inside the filter's count pass.
/// `joins` are `sys::AH_BOOL_AND` / `_OR` / `_AND_KLEENE` / `_OR_KLEENE`, folded left to right.
pub fn filter_expr(values: &Arc<DeviceArray>, terms: &[Term], joins: &[i32]) -> Result<Arc<DeviceArray>, ArrowError> {
    with_terms(terms, joins, |ctx, n, t, j| {
        let mut out = MaybeUninit::<sys::ah_array_out>::zeroed();
        let v = values.view();
        ctx.check(unsafe { sys::ah_filter_expr(ctx.raw, n, t, j, &v, out.as_mut_ptr()) })?;
        Ok(wrap(values, unsafe { out.assume_init() }, values.data_type.clone(), &[values]))
    })
}

impl FilterPredicate {
    /// the lazy form of `FilterBuilder::new(&<expression>)`: an ordinary predicate (count / filter) built from terms
    pub fn from_terms(terms: &[Term], joins: &[i32]) -> Result<Self, ArrowError> {
        with_terms(terms, joins, |ctx, n, t, j| {
            let mut raw = ptr::null_mut();
            ctx.check(unsafe { sys::ah_filter_predicate_build_expr(ctx.raw, n, t, j, &mut raw) })?;
            Ok(Self { ctx: ctx.clone(), raw, _mask: terms[0].lhs.get().0.clone() })  // (the predicate owns its selection words)
        })
    }
}

/// `TakeOptions { check_bounds }` (take.rs:388)
#[derive(Default, Clone, Copy)]
pub struct TakeOptions {
    pub check_bounds: bool,
}

/// `arrow_select::take::take` (take.rs:89)
pub fn take(values: &Arc<DeviceArray>, indices: &Arc<DeviceArray>, options: Option<TakeOptions>) -> Result<Arc<DeviceArray>, ArrowError> {
    let mut out = MaybeUninit::<sys::ah_array_out>::zeroed();
    let (v, i) = (values.view(), indices.view());
    let check = options.unwrap_or_default().check_bounds as i32;
    values.ctx.check(unsafe { sys::ah_take(values.ctx.raw, &v, &i, check, out.as_mut_ptr()) })?;
    Ok(wrap(values, unsafe { out.assume_init() }, values.data_type.clone(), &[values]))
}

// ------------------------------------------------------------------------------------------- numeric / cmp
fn arith(op: i32, lhs: &Datum, rhs: &Datum) -> Result<Arc<DeviceArray>, ArrowError> {
    let ((l, ls), (r, rs)) = (lhs.get(), rhs.get());
    let mut out = MaybeUninit::<sys::ah_array_out>::zeroed();
    let (lv, rv) = (l.view(), r.view());
    let (lt, rt) = (logical(&l.data_type)?, logical(&r.data_type)?);
    if lt.is_none() && rt.is_none() {
        l.ctx.check(unsafe { sys::ah_arith_binary(l.ctx.raw, op, &lv, ls, &rv, rs, out.as_mut_ptr()) })?;
        return Ok(wrap(l, unsafe { out.assume_init() }, l.data_type.clone(), &[]));
    }
    // temporal operands: the library applies arithmetic_op's type rules (numeric.rs:225-275) and names the result type
    let plain = |id| sys::ah_data_type { id, unit: 0, has_tz: 0, tz_offset_seconds: 0, precision: 0, scale: 0 };
    let (lt, rt) = (lt.unwrap_or(plain(lv.type_)), rt.unwrap_or(plain(rv.type_)));
    let mut ot = plain(0);
    l.ctx.check(unsafe { sys::ah_arith_with_types(l.ctx.raw, op, &lv, ls, &lt, &rv, rs, &rt, out.as_mut_ptr(), &mut ot) })?;
    let unit = |u| match u {
        sys::AH_SECOND => arrow_schema::TimeUnit::Second,
        sys::AH_MILLISECOND => arrow_schema::TimeUnit::Millisecond,
        sys::AH_MICROSECOND => arrow_schema::TimeUnit::Microsecond,
        _ => arrow_schema::TimeUnit::Nanosecond,
    };
    let result_type = match ot.id {
        sys::AH_DT_DURATION => DataType::Duration(unit(ot.unit)),
        sys::AH_DT_DECIMAL128 => DataType::Decimal128(ot.precision as u8, ot.scale as i8),  // decimal_op's Hive-rule result type
        // `array.with_timezone_opt(l.timezone())` (numeric.rs:536): the zone text of whichever side is the Timestamp
        sys::AH_DT_TIMESTAMP => match (&l.data_type, &r.data_type) {
            (t @ DataType::Timestamp(_, _), _) | (_, t @ DataType::Timestamp(_, _)) => t.clone(),
            _ => DataType::Timestamp(unit(ot.unit), None),
        },
        _ => l.data_type.clone(),
    };
    Ok(wrap(l, unsafe { out.assume_init() }, result_type, &[]))
}

macro_rules! arith_fn {
    ($($(#[$doc:meta])* $name:ident => $op:ident),* $(,)?) => {$(
        $(#[$doc])*
        pub fn $name(lhs: &Datum, rhs: &Datum) -> Result<Arc<DeviceArray>, ArrowError> { arith(sys::$op, lhs, rhs) }
    )*};
}
arith_fn! {
    /// `arrow_arith::numeric::add` (numeric.rs:36): checked for integers
    add => AH_ADD,
    /// `add_wrapping` (numeric.rs:41)
    add_wrapping => AH_ADD_WRAPPING,
    sub => AH_SUB, sub_wrapping => AH_SUB_WRAPPING, mul => AH_MUL, mul_wrapping => AH_MUL_WRAPPING,
    div => AH_DIV, rem => AH_REM,
}

/// `neg` / `neg_wrapping` (numeric.rs:103,:181)
pub fn neg(values: &Arc<DeviceArray>, wrapping: bool) -> Result<Arc<DeviceArray>, ArrowError> {
    let mut out = MaybeUninit::<sys::ah_array_out>::zeroed();
    let v = values.view();
    values.ctx.check(unsafe { sys::ah_arith_neg(values.ctx.raw, &v, wrapping as i32, out.as_mut_ptr()) })?;
    Ok(wrap(values, unsafe { out.assume_init() }, values.data_type.clone(), &[]))
}

fn compare(op: i32, lhs: &Datum, rhs: &Datum) -> Result<Arc<DeviceArray>, ArrowError> {
    let ((l, ls), (r, rs)) = (lhs.get(), rhs.get());
    let mut out = MaybeUninit::<sys::ah_array_out>::zeroed();
    let (lv, rv) = (l.view(), r.view());
    l.ctx.check(unsafe { sys::ah_compare(l.ctx.raw, op, &lv, ls, &rv, rs, out.as_mut_ptr()) })?;
    Ok(wrap(l, unsafe { out.assume_init() }, DataType::Boolean, &[]))
}

macro_rules! cmp_fn {
    ($($name:ident => $op:ident),* $(,)?) => {$(
        /// `arrow_ord::cmp` (cmp.rs:79-202): Boolean result, totalOrder for floats
        pub fn $name(lhs: &Datum, rhs: &Datum) -> Result<Arc<DeviceArray>, ArrowError> { compare(sys::$op, lhs, rhs) }
    )*};
}
cmp_fn! { eq => AH_EQ, neq => AH_NEQ, lt => AH_LT, lt_eq => AH_LT_EQ, gt => AH_GT, gt_eq => AH_GT_EQ,
          distinct => AH_DISTINCT, not_distinct => AH_NOT_DISTINCT }

// ------------------------------------------------------------------------------------------- boolean
fn boolean_binary(op: i32, l: &Arc<DeviceArray>, r: &Arc<DeviceArray>) -> Result<Arc<DeviceArray>, ArrowError> {
    let mut out = MaybeUninit::<sys::ah_array_out>::zeroed();
    let (lv, rv) = (l.view(), r.view());
    l.ctx.check(unsafe { sys::ah_boolean_binary(l.ctx.raw, op, &lv, &rv, out.as_mut_ptr()) })?;
    Ok(wrap(l, unsafe { out.assume_init() }, DataType::Boolean, &[]))
}

fn boolean_unary(op: i32, v: &Arc<DeviceArray>) -> Result<Arc<DeviceArray>, ArrowError> {
    let mut out = MaybeUninit::<sys::ah_array_out>::zeroed();
    let vv = v.view();
    v.ctx.check(unsafe { sys::ah_boolean_unary(v.ctx.raw, op, &vv, out.as_mut_ptr()) })?;
    Ok(wrap(v, unsafe { out.assume_init() }, DataType::Boolean, &[]))
}

/// `arrow_arith::boolean::and` (boolean.rs:260)
pub fn and(l: &Arc<DeviceArray>, r: &Arc<DeviceArray>) -> Result<Arc<DeviceArray>, ArrowError> { boolean_binary(sys::AH_BOOL_AND, l, r) }
/// `or` (boolean.rs:277)
pub fn or(l: &Arc<DeviceArray>, r: &Arc<DeviceArray>) -> Result<Arc<DeviceArray>, ArrowError> { boolean_binary(sys::AH_BOOL_OR, l, r) }
/// `and_kleene` (boolean.rs:60)
pub fn and_kleene(l: &Arc<DeviceArray>, r: &Arc<DeviceArray>) -> Result<Arc<DeviceArray>, ArrowError> { boolean_binary(sys::AH_BOOL_AND_KLEENE, l, r) }
/// `or_kleene` (boolean.rs:156)
pub fn or_kleene(l: &Arc<DeviceArray>, r: &Arc<DeviceArray>) -> Result<Arc<DeviceArray>, ArrowError> { boolean_binary(sys::AH_BOOL_OR_KLEENE, l, r) }
/// `not` (boolean.rs:310)
pub fn not(v: &Arc<DeviceArray>) -> Result<Arc<DeviceArray>, ArrowError> { boolean_unary(sys::AH_BOOL_NOT, v) }
/// `is_null` (boolean.rs:327)
pub fn is_null(v: &Arc<DeviceArray>) -> Result<Arc<DeviceArray>, ArrowError> { boolean_unary(sys::AH_BOOL_IS_NULL, v) }
/// `is_not_null` (boolean.rs:347)
pub fn is_not_null(v: &Arc<DeviceArray>) -> Result<Arc<DeviceArray>, ArrowError> { boolean_unary(sys::AH_BOOL_IS_NOT_NULL, v) }

// ------------------------------------------------------------------------------------------- cast / concat
/// `CastOptions { safe, .. }` (cast/mod.rs:95-111)
#[derive(Clone, Copy)]
pub struct CastOptions {
    pub safe: bool,
}

impl Default for CastOptions {
    fn default() -> Self {
        Self { safe: true }
    }
}

fn physical(ctx: &Context, t: &DataType) -> Result<sys::ah_type, ArrowError> {
    let schema = FFI_ArrowSchema::try_from(t)?;
    let mut out = 0;
    ctx.check(unsafe { sys::ah_type_from_format(ctx.raw, schema.format().as_ptr() as *const _, &mut out) })?;
    Ok(out)
}

/// `DataType` -> `ah_data_type` for the casts whose arithmetic depends on the logical type; `None` = the physical
/// type says it all.  Zones go through `arrow_array::timezone::Tz` exactly as the reference's arms do; without
/// chrono-tz that accepts fixed offsets only, which is what the C ABI carries.
fn logical(t: &DataType) -> Result<Option<sys::ah_data_type>, ArrowError> {
    use arrow_schema::TimeUnit::*;
    let unit = |u: &arrow_schema::TimeUnit| match u {
        Second => sys::AH_SECOND,
        Millisecond => sys::AH_MILLISECOND,
        Microsecond => sys::AH_MICROSECOND,
        Nanosecond => sys::AH_NANOSECOND,
    };
    let d = |id, unit, has_tz, tz_offset_seconds| sys::ah_data_type { id, unit, has_tz, tz_offset_seconds, precision: 0, scale: 0 };
    Ok(Some(match t {
        DataType::Date32 => d(sys::AH_DT_DATE32, 0, 0, 0),
        DataType::Date64 => d(sys::AH_DT_DATE64, 0, 0, 0),
        DataType::Time32(u) => d(sys::AH_DT_TIME32, unit(u), 0, 0),
        DataType::Time64(u) => d(sys::AH_DT_TIME64, unit(u), 0, 0),
        DataType::Duration(u) => d(sys::AH_DT_DURATION, unit(u), 0, 0),
        DataType::Decimal128(p, s) => sys::ah_data_type { id: sys::AH_DT_DECIMAL128, unit: 0, has_tz: 0, tz_offset_seconds: 0, precision: *p as i32, scale: *s as i32 },
        DataType::Timestamp(u, None) => d(sys::AH_DT_TIMESTAMP, unit(u), 0, 0),
        DataType::Timestamp(u, Some(tz)) => {
            use chrono::{Offset, TimeZone};
            let tz: arrow_array::timezone::Tz = tz.parse()?;
            // a fixed-offset zone has the same offset at every instant
            let secs = tz.offset_from_utc_datetime(&chrono::DateTime::UNIX_EPOCH.naive_utc()).fix().local_minus_utc();
            d(sys::AH_DT_TIMESTAMP, unit(u), 1, secs)
        }
        _ => return Ok(None),
    }))
}

/// `arrow_cast::cast_with_options` (cast/mod.rs:790): the numeric, Boolean, Utf8 / LargeUtf8 and temporal arms
pub fn cast_with_options(values: &Arc<DeviceArray>, to_type: &DataType, options: &CastOptions) -> Result<Arc<DeviceArray>, ArrowError> {
    let to = physical(&values.ctx, to_type)?;
    let mut out = MaybeUninit::<sys::ah_array_out>::zeroed();
    let v = values.view();
    match (logical(values.data_type())?, logical(to_type)?) {
        (None, None) => values.ctx.check(unsafe { sys::ah_cast(values.ctx.raw, &v, to, options.safe as i32, out.as_mut_ptr()) })?,
        (f, t) => {
            let plain = |id| sys::ah_data_type { id, unit: 0, has_tz: 0, tz_offset_seconds: 0, precision: 0, scale: 0 };
            let (f, t) = (f.unwrap_or(plain(v.type_)), t.unwrap_or(plain(to)));
            values.ctx.check(unsafe { sys::ah_cast_with_types(values.ctx.raw, &v, &f, &t, options.safe as i32, out.as_mut_ptr()) })?
        }
    }
    Ok(wrap(values, unsafe { out.assume_init() }, to_type.clone(), &[]))
}

/// `arrow_cast::cast` (cast/mod.rs:347)
pub fn cast(values: &Arc<DeviceArray>, to_type: &DataType) -> Result<Arc<DeviceArray>, ArrowError> {
    cast_with_options(values, to_type, &CastOptions::default())
}

/// `arrow_select::concat::concat` (concat.rs:495); also the multi-GPU reassembly primitive
pub fn concat(arrays: &[&Arc<DeviceArray>]) -> Result<Arc<DeviceArray>, ArrowError> {
    let first = arrays.first().ok_or_else(|| ArrowError::InvalidArgumentError("concat requires input of at least one array".into()))?;
    let views: Vec<sys::ah_array_view> = arrays.iter().map(|a| a.view()).collect();
    let mut out = MaybeUninit::<sys::ah_array_out>::zeroed();
    first.ctx.check(unsafe { sys::ah_concat(first.ctx.raw, views.len() as i32, views.as_ptr(), out.as_mut_ptr()) })?;
    Ok(wrap(first, unsafe { out.assume_init() }, first.data_type.clone(), &[]))
}

/// `arrow_select::window::shift` (window.rs:56): positive offsets shift right, vacated slots are null
pub fn shift(values: &Arc<DeviceArray>, offset: i64) -> Result<Arc<DeviceArray>, ArrowError> {
    let mut out = MaybeUninit::<sys::ah_array_out>::zeroed();
    let v = values.view();
    values.ctx.check(unsafe { sys::ah_shift(values.ctx.raw, &v, offset, out.as_mut_ptr()) })?;
    Ok(wrap(values, unsafe { out.assume_init() }, values.data_type.clone(), &[values]))
}

/// `arrow_ord::rank::rank` (rank.rs:58), the ranks left on the device as a `UInt32` array (the reference's
/// `Vec<u32>` is `to_host()` of it)
pub fn rank(values: &Arc<DeviceArray>, options: Option<SortOptions>) -> Result<Arc<DeviceArray>, ArrowError> {
    let o = options.unwrap_or_default();
    let mut out = MaybeUninit::<sys::ah_array_out>::zeroed();
    let v = values.view();
    values.ctx.check(unsafe { sys::ah_rank(values.ctx.raw, &v, o.descending as i32, o.nulls_first as i32, out.as_mut_ptr()) })?;
    Ok(wrap(values, unsafe { out.assume_init() }, DataType::UInt32, &[]))
}


// ------------------------------------------------------------------------------------------- BatchCoalescer
/// `arrow_select::coalesce::BatchCoalescer` (coalesce.rs:148) over fixed-width device columns: the state machine
/// is native (`ah_coalescer_*`); filtered pushes scatter straight into the in-progress batch and never wait for
/// the GPU except for the predicate's count.
pub struct BatchCoalescer {
    ctx: Arc<Context>,
    raw: *mut sys::ah_coalescer,
    types: Vec<DataType>,
    /// input batches that were passed through untouched (large-batch bypass), by tag, until they are popped
    bypassed: std::collections::HashMap<u64, Vec<Arc<DeviceArray>>>,
    next_tag: u64,
}

impl BatchCoalescer {
    /// `BatchCoalescer::new(schema, target_batch_size)` (coalesce.rs:167)
    pub fn new(ctx: &Arc<Context>, types: &[DataType], target_batch_size: usize) -> Result<Self, ArrowError> {
        let phys: Vec<sys::ah_type> = types.iter().map(|t| physical(ctx, t)).collect::<Result<_, _>>()?;
        let mut raw = ptr::null_mut();
        ctx.check(unsafe {
            sys::ah_coalescer_create(ctx.raw, phys.len() as i32, phys.as_ptr(), target_batch_size as i64, &mut raw)
        })?;
        Ok(Self { ctx: ctx.clone(), raw, types: types.to_vec(), bypassed: Default::default(), next_tag: 1 })
    }

    /// `with_biggest_coalesce_batch_size` (coalesce.rs:196)
    pub fn with_biggest_coalesce_batch_size(self, limit: Option<usize>) -> Self {
        unsafe { sys::ah_coalescer_set_biggest_coalesce_batch_size(self.raw, limit.map_or(-1, |l| l as i64)) };
        self
    }

    fn push(&mut self, columns: &[Arc<DeviceArray>], filter: Option<&Arc<DeviceArray>>) -> Result<(), ArrowError> {
        let views: Vec<sys::ah_array_view> = columns.iter().map(|c| c.view()).collect();
        let rows = columns.first().map_or(0, |c| c.len()) as i64;
        let (tag, mut bypass) = (self.next_tag, 0i32);
        self.next_tag += 1;
        let st = match filter {
            None => unsafe { sys::ah_coalescer_push_batch(self.ctx.raw, self.raw, views.as_ptr(), rows, tag, &mut bypass) },
            Some(f) => {
                let fv = f.view();
                unsafe {
                    sys::ah_coalescer_push_batch_with_filter(self.ctx.raw, self.raw, views.as_ptr(), rows, &fv, tag, &mut bypass)
                }
            }
        };
        self.ctx.check(st)?;
        if bypass != 0 {
            self.bypassed.insert(tag, columns.to_vec());
        }
        Ok(())
    }

    /// `push_batch` (coalesce.rs:296)
    pub fn push_batch(&mut self, columns: &[Arc<DeviceArray>]) -> Result<(), ArrowError> {
        self.push(columns, None)
    }

    /// `push_batch_with_filter` (coalesce.rs:229)
    pub fn push_batch_with_filter(&mut self, columns: &[Arc<DeviceArray>], filter: &Arc<DeviceArray>) -> Result<(), ArrowError> {
        self.push(columns, Some(filter))
    }

    /// `push_batch_with_indices` (coalesce.rs:289): `take_record_batch(batch, indices)` into the in-progress batch
    pub fn push_batch_with_indices(&mut self, columns: &[Arc<DeviceArray>], indices: &Arc<DeviceArray>) -> Result<(), ArrowError> {
        let views: Vec<sys::ah_array_view> = columns.iter().map(|c| c.view()).collect();
        let rows = columns.first().map_or(0, |c| c.len()) as i64;
        let iv = indices.view();
        self.ctx.check(unsafe { sys::ah_coalescer_push_batch_with_indices(self.ctx.raw, self.raw, views.as_ptr(), rows, &iv) })
    }

    /// `finish_buffered_batch` (coalesce.rs:547); does not wait: null counts are looked at when the batch is fetched
    pub fn finish_buffered_batch(&mut self) -> Result<(), ArrowError> {
        self.ctx.check(unsafe { sys::ah_coalescer_finish_buffered_batch(self.ctx.raw, self.raw) })
    }

    pub fn has_completed_batch(&self) -> bool {
        unsafe { sys::ah_coalescer_completed_count(self.raw) > 0 }
    }

    pub fn get_buffered_rows(&self) -> usize {
        unsafe { sys::ah_coalescer_buffered_rows(self.raw) as usize }
    }

    /// `next_completed_batch` (coalesce.rs:566): the columns of the next exact-size batch
    pub fn next_completed_batch(&mut self) -> Result<Option<Vec<Arc<DeviceArray>>>, ArrowError> {
        let n = self.types.len();
        let mut outs: Vec<sys::ah_array_out> = (0..n).map(|_| unsafe { std::mem::zeroed() }).collect();
        let (mut rows, mut tag) = (0i64, 0u64);
        self.ctx.check(unsafe {
            sys::ah_coalescer_next_completed_batch(self.ctx.raw, self.raw, outs.as_mut_ptr(), &mut rows, &mut tag)
        })?;
        if rows < 0 {
            return Ok(None);
        }
        if tag != 0 {
            return Ok(self.bypassed.remove(&tag)); // the caller's own batch, untouched
        }
        Ok(Some(
            outs.into_iter()
                .zip(self.types.iter())
                .map(|(out, dt)| Arc::new(DeviceArray { ctx: self.ctx.clone(), out, data_type: dt.clone(), _keep: vec![] }))
                .collect(),
        ))
    }
}

impl Drop for BatchCoalescer {
    fn drop(&mut self) {
        unsafe { sys::ah_coalescer_destroy(self.ctx.raw, self.raw) }
    }
}

// ------------------------------------------------------------------------------------------- multi-GPU exchange
/// One rank of the row-sharded exchange (one process per GPU).  libarrow_hip.so binds RCCL itself; the host only
/// ships rank 0's id to the other ranks (its control plane, a file, MPI ...).  INTEGRATION.md §2d.
pub struct Comm {
    ctx: Arc<Context>,
    raw: *mut sys::ah_comm,
}

impl Comm {
    /// rank 0: the 128 bytes every rank passes to [`Comm::new`]
    pub fn unique_id(ctx: &Context) -> Result<[u8; sys::AH_COMM_ID_BYTES], ArrowError> {
        let mut id = [0u8; sys::AH_COMM_ID_BYTES];
        ctx.check(unsafe { sys::ah_comm_unique_id(ctx.raw, id.as_mut_ptr()) })?;
        Ok(id)
    }

    /// collective over all ranks (`ncclCommInitRank`)
    pub fn new(ctx: &Arc<Context>, rank: i32, world: i32, id: &[u8; sys::AH_COMM_ID_BYTES]) -> Result<Self, ArrowError> {
        let mut raw = ptr::null_mut();
        ctx.check(unsafe { sys::ah_comm_create(ctx.raw, rank, world, id.as_ptr(), &mut raw) })?;
        Ok(Self { ctx: ctx.clone(), raw })
    }

    /// `concat(rank 0's array, rank 1's array, ...)` on every rank (arrow-select/src/concat.rs:495 across ranks)
    pub fn all_gatherv(&self, local: &Arc<DeviceArray>) -> Result<Arc<DeviceArray>, ArrowError> {
        let mut out = MaybeUninit::<sys::ah_array_out>::zeroed();
        let v = local.view();
        self.ctx.check(unsafe { sys::ah_all_gatherv(self.ctx.raw, self.raw, &v, out.as_mut_ptr(), ptr::null_mut()) })?;
        Ok(Arc::new(DeviceArray { ctx: self.ctx.clone(), out: unsafe { out.assume_init() }, data_type: local.data_type.clone(), _keep: vec![] }))
    }

    /// the columns of a RecordBatch shard in ONE count exchange + ONE grouped send / recv (`concat_batches`, concat.rs:607)
    pub fn all_gather_columns(&self, shard: &[Arc<DeviceArray>]) -> Result<Vec<Arc<DeviceArray>>, ArrowError> {
        let views: Vec<sys::ah_array_view> = shard.iter().map(|c| c.view()).collect();
        let mut outs: Vec<sys::ah_array_out> = (0..shard.len()).map(|_| unsafe { std::mem::zeroed() }).collect();
        self.ctx.check(unsafe {
            sys::ah_all_gather_columns(self.ctx.raw, self.raw, views.len() as i32, views.as_ptr(), outs.as_mut_ptr(), ptr::null_mut())
        })?;
        Ok(outs
            .into_iter()
            .zip(shard.iter())
            .map(|(out, c)| Arc::new(DeviceArray { ctx: self.ctx.clone(), out, data_type: c.data_type.clone(), _keep: vec![] }))
            .collect())
    }

    pub fn barrier(&self) -> Result<(), ArrowError> {
        self.ctx.check(unsafe { sys::ah_comm_barrier(self.ctx.raw, self.raw) })
    }
}

impl Drop for Comm {
    fn drop(&mut self) {
        unsafe { sys::ah_comm_destroy(self.ctx.raw, self.raw) }
    }
}
