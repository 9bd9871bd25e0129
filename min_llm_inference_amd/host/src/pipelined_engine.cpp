#include "pipelined_engine.h"

#include <algorithm>
#include <cstring>
#include <stdexcept>
#include <vector>

#include "constants.h"
#include "runtime.h"
#include "throughput_counter.h"

namespace {

struct Range {  // roctx range, closed on scope exit
    explicit Range(const char* name) { mli::runtime::range_push(name); }
    ~Range() { mli::runtime::range_pop(); }
};

struct MarkerHandle {
    mli::mem::Marker* m;
    MarkerHandle() : m(mli::mem::create_marker()) {}
    ~MarkerHandle() { mli::mem::destroy_marker(m); }
    MarkerHandle(const MarkerHandle&) = delete;
    MarkerHandle& operator=(const MarkerHandle&) = delete;
};

}  // namespace

long long run_paged_engine_pipelined(ItemStorage& item_storage, ProcessingStorage& processing_storage,
                                     MemoryBlockManager& pool, PagedAttentionsManager& pages, size_t n_batch_size,
                                     size_t n_sequence, const PagedForward& forward, int n_forward_rounds) {
    // Every in-flight row has up to R tokens in flight and is about to produce R more, so the page bookkeeping is asked
    // for tokens + 2 R positions; its rule "a row needs at most one more page per pass" holds while 2 R <= PAGE_BLOCK_SIZE.
    const int R = n_forward_rounds;
    if (R < 1 || 2 * R > PAGE_BLOCK_SIZE)
        throw std::runtime_error("the pipelined engine serves 1 <= n_forward_rounds <= PAGE_BLOCK_SIZE / 2");
    if (pages.length_reset_quirk())
        throw std::runtime_error("the pipelined engine does not reproduce the reference's length-reset quirk");
    const int B = static_cast<int>(n_batch_size), S = static_cast<int>(n_sequence);
    // The loop's own buffers are sync-flavour whatever the process default (memory.h): its per-slot updates are the
    // stream-ordered, non-blocking copies of copy_async, which the async flavour -- every copy on the transfer stream,
    // every data() a wait -- does not offer.
    constexpr TensorDataType kSync = TensorDataType::SYNC_ALLOCATE;
    TensorInt inp_device({n_batch_size, n_sequence}, DeviceType::DEVICE, kSync), inp_host({n_batch_size, n_sequence}, DeviceType::HOST, kSync);
    TensorInt lengths_device({n_batch_size}, DeviceType::DEVICE, kSync), lengths_host({n_batch_size}, DeviceType::HOST, kSync);
    TensorInt new_idx_device({n_batch_size}, DeviceType::DEVICE, kSync);
    // staging for the new-row indices of forward(k) is reused for forward(k+2): by then the marker of step k+1,
    // recorded after forward(k+1) was queued, has been waited for, so the copy that fed forward(k) has executed
    TensorInt new_idx_host[2] = {TensorInt({n_batch_size}, DeviceType::HOST, kSync), TensorInt({n_batch_size}, DeviceType::HOST, kSync)};
    const size_t n_rounds = static_cast<size_t>(R);
    TensorInt result_device({n_batch_size, n_rounds}, DeviceType::DEVICE, kSync), result_host({n_batch_size, n_rounds}, DeviceType::HOST, kSync);
    MarkerHandle marker;

    // first_step[b] = index of the first forward this occupant of slot b takes part in; -1 = slot empty
    std::vector<long long> first_step(n_batch_size, -1);
    std::vector<long long> idx;
    std::vector<int> val;

    std::memset(lengths_host.data(), 0, n_batch_size * sizeof(int));
    lengths_device.copy_from(lengths_host);
    get_global_throughput_counter().start_record();

    // admission for forward `step`: host decisions, then per-slot, stream-ordered, non-blocking device updates
    auto admit = [&](long long step) {
        Range r("insert_new_items");
        TensorInt& staging = new_idx_host[step & 1];
        const bool nothing_in_flight = processing_storage.size() == 0;  // then every slot and every page is free
        PagedAdmission adm = admit_new_items(inp_host.data(), lengths_host.data(), staging.data(), B, S, item_storage,
                                             processing_storage, pool, pages, R);
        if (nothing_in_flight && adm.slots.empty() && item_storage.new_count() > 0) {
            // the whole pool cannot hold the head item: an error, not the endless loop of the reference's engine
            mli::runtime::synchronize();
            throw std::runtime_error("paged engine: the page pool is too small for the next queued item");
        }
        if (adm.slots.empty()) {
            pages.maybe_flush_changes();
            return 0;
        }
        idx.clear();
        val.clear();
        for (int slot : adm.slots) {
            first_step[slot] = step;
            idx.push_back(slot);
            val.push_back(lengths_host.data()[slot]);
        }
        if (adm.slots.size() <= 8) {
            for (int slot : adm.slots)
                inp_device.copy_range_from_async(inp_host, static_cast<size_t>(slot) * S, static_cast<size_t>(lengths_host.data()[slot]));
        } else {
            // one span first..last admitted row: rows in between are unchanged (in flight) or free (not read until
            // their own admission uploads them again)
            const int lo = adm.slots.front(), hi = adm.slots.back();
            inp_device.copy_range_from_async(inp_host, static_cast<size_t>(lo) * S,
                                             static_cast<size_t>(hi - lo) * S + lengths_host.data()[hi]);
        }
        lengths_device.scatter_from_host(idx.data(), val.data(), idx.size());
        new_idx_device.copy_range_from_async(staging, 0, adm.slots.size());
        pages.maybe_flush_changes();
        return static_cast<int>(adm.slots.size());
    };

    // E. result(step): wait for its copy, append the tokens, retire finished rows, hand their pages back
    auto process_result = [&](long long step) {
        std::vector<int> finished;
        {
            Range r("process_decoder_result");
            mli::mem::wait_marker(marker.m);
            const int* tokens = result_host.data();
            int appended = 0;
            for (int b = 0; b < B; ++b) {
                if (first_step[b] < 0 || first_step[b] > step) continue;  // empty, or admitted after forward(step)
                IdTokensPair& item = processing_storage.get_token(b);
                for (int r = 0; r < R; ++r) {  // one token per round until the row finishes (later rounds: EMPTY)
                    const int tok = tokens[static_cast<size_t>(b) * R + r];
                    if (tok == EMPTY_ROW_TOKEN_ID) throw std::runtime_error("pipelined engine: in-flight row reported empty");
                    append_token_to_id_string_pair(item, tok);
                    ++appended;
                    if (tok == EOF_TOKEN_ID || static_cast<int>(item.second.size()) >= S) {
                        processing_storage.move_to_finished(b, item_storage);
                        first_step[b] = -1;
                        lengths_host.data()[b] = 0;  // the decoder zeroed the device length already
                        finished.push_back(b);
                        break;
                    }
                }
            }
            get_global_throughput_counter().add_record_if_recording(appended);
        }
        return finished;
    };
    // rows the bookkeeping has just preempted: their device length is zeroed before the next forward
    auto zero_preempted_rows = [&]() {
        idx.clear();
        val.clear();
        for (int b = 0; b < B; ++b) {
            if (first_step[b] >= 0 && !processing_storage.batch_id_processing(b)) {
                first_step[b] = -1;
                lengths_host.data()[b] = 0;
                idx.push_back(b);
                val.push_back(0);
            }
        }
        if (!idx.empty()) lengths_device.scatter_from_host(idx.data(), val.data(), idx.size());
    };

    long long step = 0;
    int n_new = admit(0);
    {
        Range r("forward");
        forward(inp_device, lengths_device, new_idx_device, result_device, n_new);
    }
    while (true) {
        // A. result(step) starts travelling as soon as forward(step) is done
        result_host.copy_range_from_async(result_device, 0, n_batch_size * n_rounds);
        mli::mem::record_marker(marker.m);

        // B. pages for forward(step + 1): every in-flight row has up to R tokens in flight (one per round of
        //    forward(step)) and forward(step + 1) appends up to R more, so it needs room for tokens + 2 R positions
        //    (the reference's rule with the in-flight tokens counted); a dry pool preempts from the tail of the
        //    admission list, and the victim's device length is zeroed before forward(step + 1).
        //    The look-ahead must not cost the LAST row its place: it may be finishing in forward(step) (EOF, or its
        //    last token landing on a page boundary), in which case it never needs the page -- the sequential loop, which
        //    asks for tokens + R after the result, completes such a workload (pools smaller than the table width).
        //    So when the only row in flight cannot get its look-ahead page, this iteration runs in the reference's
        //    order: result(step) first, then the sequential rule decides (and preempts, if the row really needs it).
        bool result_done = false;
        {
            Range r("allocate_or_free_memory_blocks_if_needed");
            bool last_row_short = false;
            allocate_or_free_memory_blocks_if_needed(pages, pool, processing_storage, item_storage, {}, /*rounds=*/2 * R,
                                                     &last_row_short);
            if (last_row_short) {
                const std::vector<int> finished = process_result(step);
                allocate_or_free_memory_blocks_if_needed(pages, pool, processing_storage, item_storage, finished, /*rounds=*/R);
                result_done = true;
            }
            zero_preempted_rows();
        }

        // C. + D. admission into the slots known to be free, then the next forward
        //    (nothing in flight, nothing admitted, items queued: admit() reports "pool too small", as the sequential
        //    loops do after their admission refuses the row -- inferencer.cpp: throw_if_stuck)
        n_new = admit(step + 1);
        {
            Range r("forward");
            forward(inp_device, lengths_device, new_idx_device, result_device, n_new);
        }

        // E. result(step), unless B already needed it
        if (!result_done) {
            const std::vector<int> finished = process_result(step);
            // pages of finished rows go back; with R rounds and the tokens just appended this asks for no more than B
            // already provided, so nothing grows here
            if (!finished.empty())
                allocate_or_free_memory_blocks_if_needed(pages, pool, processing_storage, item_storage, finished, /*rounds=*/R);
        }
        ++step;
        if (is_done(item_storage, processing_storage)) break;
    }
    mli::runtime::synchronize();  // forward(step) is still in flight (every row empty): drain before the tensors go
    return step + 1;
}
