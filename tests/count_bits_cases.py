"""Buffer::count_set_bits_offset's inline vectors (arrow-buffer/src/buffer/immutable.rs:807-890: test_count_bits,
test_count_bits_slice, test_count_bits_offset_slice) — the popcount under BooleanArray::true_count and every null count
(SURVEY §8a row F1).  TEST INFRASTRUCTURE.  (bytes, byte offset of the slice, bit offset, bit length, expected)"""
CASES = [
    # test_count_bits (:808-820)
    ([0b00000000], 0, 0, 8, 0), ([0b11111111], 0, 0, 8, 8), ([0b00001101], 0, 0, 8, 3),
    ([0b01001001, 0b01010010], 0, 0, 16, 6), ([0b11111111, 0b11111111], 0, 0, 16, 16),
    # test_count_bits_slice (:823-853): Buffer::slice(n) = a byte offset
    ([0b11111111, 0b00000000], 1, 0, 8, 0), ([0b11111111, 0b11111111], 1, 0, 8, 8), ([0b11111111, 0b11111111, 0b00001101], 2, 0, 8, 3),
    ([0b11111111, 0b01001001, 0b01010010], 1, 0, 16, 6), ([0b11111111] * 4, 2, 0, 16, 16),
    # test_count_bits_offset_slice (:856-890)
    ([0b11111111], 0, 0, 8, 8), ([0b11111111], 0, 0, 3, 3), ([0b11111111], 0, 3, 5, 5), ([0b11111111], 0, 3, 1, 1),
    ([0b11111111], 0, 8, 0, 0), ([0b01010101], 0, 0, 3, 2), ([0b11111111, 0b11111111], 0, 0, 16, 16),
    ([0b11111111, 0b11111111], 0, 0, 10, 10), ([0b11111111, 0b11111111], 0, 3, 10, 10), ([0b11111111, 0b11111111], 0, 8, 8, 8),
    ([0b11111111, 0b11111111], 0, 11, 5, 5), ([0b11111111, 0b11111111], 0, 16, 0, 0), ([0b01101101, 0b10101010], 0, 7, 5, 2),
    ([0b01101101, 0b10101010], 0, 7, 9, 4),
]
