// device_spmv.hpp -- header-only C++ shim with the reference's spelling over the
// C ABI of include/mspmv.h, so that a driver written against
//     cub::DeviceSpmv::CsrMV(d_temp_storage, temp_storage_bytes, d_values,
//         d_row_offsets, d_column_indices, d_vector_x, d_vector_y,
//         num_rows, num_cols, num_nonzeros, stream, debug_synchronous)
// (reference cub/device/device_spmv.cuh:129-145) keeps its source shape:
// replace `cub::` by `mspmv::` and cudaStream_t by hipStream_t.
#pragma once

#include <hip/hip_runtime.h>

#include "../../include/mspmv.h"

namespace mspmv {

struct DeviceSpmv {
    static hipError_t CsrMV(void *d_temp_storage, size_t &temp_storage_bytes, const float *d_values,
                            const int *d_row_offsets, const int *d_column_indices, const float *d_vector_x,
                            float *d_vector_y, int num_rows, int num_cols, int num_nonzeros, hipStream_t stream = 0,
                            bool debug_synchronous = false)
    {
        return (hipError_t) mspmv_csrmv_f32(d_temp_storage, &temp_storage_bytes, d_values, d_row_offsets,
                                            d_column_indices, d_vector_x, d_vector_y, num_rows, num_cols,
                                            num_nonzeros, (mspmv_stream_t) stream, debug_synchronous ? 1 : 0);
    }

    static hipError_t CsrMV(void *d_temp_storage, size_t &temp_storage_bytes, const double *d_values,
                            const int *d_row_offsets, const int *d_column_indices, const double *d_vector_x,
                            double *d_vector_y, int num_rows, int num_cols, int num_nonzeros, hipStream_t stream = 0,
                            bool debug_synchronous = false)
    {
        return (hipError_t) mspmv_csrmv_f64(d_temp_storage, &temp_storage_bytes, d_values, d_row_offsets,
                                            d_column_indices, d_vector_x, d_vector_y, num_rows, num_cols,
                                            num_nonzeros, (mspmv_stream_t) stream, debug_synchronous ? 1 : 0);
    }

    // y = alpha*A*x + beta*y (extension; the reference's --alpha/--beta are honoured only by its SpmvGold)
    static hipError_t CsrMV(void *d_temp_storage, size_t &temp_storage_bytes, const float *d_values,
                            const int *d_row_offsets, const int *d_column_indices, const float *d_vector_x,
                            float *d_vector_y, int num_rows, int num_cols, int num_nonzeros, float alpha, float beta,
                            hipStream_t stream = 0, bool debug_synchronous = false)
    {
        return (hipError_t) mspmv_csrmv_axpby_f32(d_temp_storage, &temp_storage_bytes, d_values, d_row_offsets,
                                                  d_column_indices, d_vector_x, d_vector_y, num_rows, num_cols,
                                                  num_nonzeros, alpha, beta, (mspmv_stream_t) stream,
                                                  debug_synchronous ? 1 : 0);
    }

    static hipError_t CsrMV(void *d_temp_storage, size_t &temp_storage_bytes, const double *d_values,
                            const int *d_row_offsets, const int *d_column_indices, const double *d_vector_x,
                            double *d_vector_y, int num_rows, int num_cols, int num_nonzeros, double alpha, double beta,
                            hipStream_t stream = 0, bool debug_synchronous = false)
    {
        return (hipError_t) mspmv_csrmv_axpby_f64(d_temp_storage, &temp_storage_bytes, d_values, d_row_offsets,
                                                  d_column_indices, d_vector_x, d_vector_y, num_rows, num_cols,
                                                  num_nonzeros, alpha, beta, (mspmv_stream_t) stream,
                                                  debug_synchronous ? 1 : 0);
    }

    // ---- iterated SpMV on one matrix (extension): find the tile coordinates once ...
    template <typename ValueT>
    static hipError_t CsrMVPrepare(void *d_temp_storage, size_t &temp_storage_bytes, const int *d_row_offsets, int num_rows,
                                   int num_nonzeros, hipStream_t stream = 0, bool debug_synchronous = false)
    {
        return (hipError_t) mspmv_csrmv_prepare(d_temp_storage, &temp_storage_bytes, d_row_offsets, num_rows, num_nonzeros,
                                                (int) sizeof(ValueT), (mspmv_stream_t) stream, debug_synchronous ? 1 : 0);
    }
    // ... then y = alpha*A*x + beta*y without the coordinate pass (d_temp_storage as left by CsrMVPrepare)
    static hipError_t CsrMVPrepared(void *d_temp_storage, size_t &temp_storage_bytes, const float *d_values,
                                    const int *d_row_offsets, const int *d_column_indices, const float *d_vector_x,
                                    float *d_vector_y, int num_rows, int num_cols, int num_nonzeros, float alpha = 1.f,
                                    float beta = 0.f, hipStream_t stream = 0, bool debug_synchronous = false)
    {
        return (hipError_t) mspmv_csrmv_prepared_f32(d_temp_storage, &temp_storage_bytes, d_values, d_row_offsets,
                                                     d_column_indices, d_vector_x, d_vector_y, num_rows, num_cols,
                                                     num_nonzeros, alpha, beta, (mspmv_stream_t) stream,
                                                     debug_synchronous ? 1 : 0);
    }
    static hipError_t CsrMVPrepared(void *d_temp_storage, size_t &temp_storage_bytes, const double *d_values,
                                    const int *d_row_offsets, const int *d_column_indices, const double *d_vector_x,
                                    double *d_vector_y, int num_rows, int num_cols, int num_nonzeros, double alpha = 1.0,
                                    double beta = 0.0, hipStream_t stream = 0, bool debug_synchronous = false)
    {
        return (hipError_t) mspmv_csrmv_prepared_f64(d_temp_storage, &temp_storage_bytes, d_values, d_row_offsets,
                                                     d_column_indices, d_vector_x, d_vector_y, num_rows, num_cols,
                                                     num_nonzeros, alpha, beta, (mspmv_stream_t) stream,
                                                     debug_synchronous ? 1 : 0);
    }

    // ---- prepared band-major plan (extension, opt-in): build once, multiply many times
    template <typename ValueT>
    static hipError_t PlanSize(int num_rows, int num_cols, int num_nonzeros, int bands, size_t &plan_bytes, int &bands_used)
    {
        int32_t used = 0;
        const hipError_t e = (hipError_t) mspmv_csrmv_plan_size(num_rows, num_cols, num_nonzeros, (int) sizeof(ValueT), bands, &plan_bytes, &used);
        bands_used = used;
        return e;
    }
    static hipError_t PlanBuild(void *d_plan, size_t plan_bytes, const float *d_values, const int *d_row_offsets,
                                const int *d_column_indices, int num_rows, int num_cols, int num_nonzeros, int bands,
                                hipStream_t stream = 0, bool debug_synchronous = false)
    {
        return (hipError_t) mspmv_csrmv_plan_build_f32(d_plan, plan_bytes, d_values, d_row_offsets, d_column_indices, num_rows, num_cols,
                                                       num_nonzeros, bands, (mspmv_stream_t) stream, debug_synchronous ? 1 : 0);
    }
    static hipError_t PlanBuild(void *d_plan, size_t plan_bytes, const double *d_values, const int *d_row_offsets,
                                const int *d_column_indices, int num_rows, int num_cols, int num_nonzeros, int bands,
                                hipStream_t stream = 0, bool debug_synchronous = false)
    {
        return (hipError_t) mspmv_csrmv_plan_build_f64(d_plan, plan_bytes, d_values, d_row_offsets, d_column_indices, num_rows, num_cols,
                                                       num_nonzeros, bands, (mspmv_stream_t) stream, debug_synchronous ? 1 : 0);
    }
    static hipError_t PlanApply(void *d_plan, size_t plan_bytes, const float *d_vector_x, float *d_vector_y, int num_rows, int num_cols,
                                int num_nonzeros, int bands, float alpha = 1.f, float beta = 0.f, hipStream_t stream = 0,
                                bool debug_synchronous = false)
    {
        return (hipError_t) mspmv_csrmv_plan_apply_f32(d_plan, plan_bytes, d_vector_x, d_vector_y, num_rows, num_cols, num_nonzeros, bands,
                                                       alpha, beta, (mspmv_stream_t) stream, debug_synchronous ? 1 : 0);
    }
    static hipError_t PlanApply(void *d_plan, size_t plan_bytes, const double *d_vector_x, double *d_vector_y, int num_rows, int num_cols,
                                int num_nonzeros, int bands, double alpha = 1.0, double beta = 0.0, hipStream_t stream = 0,
                                bool debug_synchronous = false)
    {
        return (hipError_t) mspmv_csrmv_plan_apply_f64(d_plan, plan_bytes, d_vector_x, d_vector_y, num_rows, num_cols, num_nonzeros, bands,
                                                       alpha, beta, (mspmv_stream_t) stream, debug_synchronous ? 1 : 0);
    }
};

}  // namespace mspmv
