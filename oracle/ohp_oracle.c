/*
 * ohp_oracle.c -- CPU restatement of ohPipeline's PCM hot path.  TEST INFRASTRUCTURE ONLY
 * (see ohp_oracle.h for the parity status of every row).  Plain C, no dependencies.
 * Citations are file:line relative to /root/reference.
 */
#include "ohp_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------
 * a7 table.  RampArray.h:7-74 holds 512 Q15 multipliers described as "a ramp down curve over
 * 0 to -60dB".  They satisfy, for every i, kRampArray[i] == min(32767, round(32768*(1-i/512)^2.5))
 * (tests/test_oracle_ramp_table.py checks all 512 against tests/golden/ramp_table_q15.json).
 * With n = 512-i:  32768*(n/512)^2.5 = sqrt(n^5 / 2^15); rounding done exactly in integers.
 * ---------------------------------------------------------------------------------------- */
static uint16_t g_ramp_table[OHP_RAMP_TABLE_COUNT];
static int g_ramp_table_ready = 0;

static void build_ramp_table(void)
{
    for (uint32_t i = 0; i < OHP_RAMP_TABLE_COUNT; i++) {
        const uint64_t n = 512u - i;
        const uint64_t n5 = n * n * n * n * n;          /* <= 2^45 */
        /* v = floor(sqrt(n5/2^15) + 0.5)  <=>  largest v with (2v-1)^2 * 2^13 <= n5 */
        uint64_t v = (uint64_t)floor(sqrt((double)n5 / 32768.0) + 0.5);
        while (v > 0 && (2 * v - 1) * (2 * v - 1) * 8192u > n5) v--;
        while ((2 * v + 1) * (2 * v + 1) * 8192u <= n5) v++;
        if (v > 32767u) v = 32767u;
        g_ramp_table[i] = (uint16_t)v;
    }
    g_ramp_table_ready = 1;
}

const uint16_t* ohp_ramp_table(void)
{
    if (!g_ramp_table_ready) build_ramp_table();
    return g_ramp_table;
}

/* ------------------------------------------------------------------------------------------
 * a1  DecodedAudio::ConstructPcm (Msg.cpp:347-368) + CopyToBigEndian16/24/32 (Msg.cpp:380-408)
 * ---------------------------------------------------------------------------------------- */
int ohp_construct_pcm(const uint8_t* src, uint32_t bytes, uint32_t bit_depth, int endian, uint8_t* dst)
{
    if ((bit_depth & 7) != 0) return OHP_ERR_ASSERT;                 /* Msg.cpp:349 */
    if (bit_depth == 0 || bytes % (bit_depth / 8) != 0) return OHP_ERR_ASSERT; /* Msg.cpp:350 */
    if (bytes > OHP_MAX_BYTES) return OHP_ERR_ASSERT;                /* Bws<kMaxBytes>, Msg.h:134 */
    if (endian == OHP_ENDIAN_BIG || bit_depth == 8) {
        memcpy(dst, src, bytes);
    } else if (bit_depth == 16) {
        for (uint32_t i = 0; i < bytes; i += 2) { *dst++ = src[i + 1]; *dst++ = src[i]; }
    } else if (bit_depth == 24) {
        for (uint32_t i = 0; i < bytes; i += 3) { *dst++ = src[i + 2]; *dst++ = src[i + 1]; *dst++ = src[i]; }
    } else if (bit_depth == 32) {
        for (uint32_t i = 0; i < bytes; i += 4) { *dst++ = src[i + 3]; *dst++ = src[i + 2]; *dst++ = src[i + 1]; *dst++ = src[i]; }
    } else {
        return OHP_ERR_ASSERT;                                        /* Msg.cpp:364-366 */
    }
    return OHP_OK;
}

/* ------------------------------------------------------------------------------------------
 * a2  Jiffies (Msg.cpp:424-527, Msg.h:193-237)
 * ---------------------------------------------------------------------------------------- */
int ohp_jiffies_per_sample(uint32_t sample_rate)
{
    switch (sample_rate) {                                            /* Msg.cpp:424-474 */
    case 7350: case 8000: case 11025: case 12000: case 14700: case 16000: case 22050:
    case 24000: case 29400: case 32000: case 44100: case 48000: case 88200: case 96000:
    case 176400: case 192000: case 352800: case 384000:
    case 2822400: case 5644800: case 11289600:
        return (int)(OHP_JIFFIES_PER_SEC / sample_rate);              /* Msg.h:213-233 */
    default:
        return OHP_ERR_SAMPLE_RATE;
    }
}

uint32_t ohp_jiffies_to_bytes_sample_block(uint32_t* jiffies, uint32_t jps, uint32_t channels, uint32_t bits, uint32_t samples_per_block)
{                                                                     /* Msg.cpp:481-489 */
    *jiffies -= *jiffies % (jps * samples_per_block);
    const uint32_t num_samples = *jiffies / jps;
    const uint32_t num_subsamples = num_samples * channels;
    return ((num_subsamples * bits) + 7) / 8;
}

uint32_t ohp_jiffies_to_bytes(uint32_t* jiffies, uint32_t jps, uint32_t channels, uint32_t bits)
{                                                                     /* Msg.cpp:476-479 */
    return ohp_jiffies_to_bytes_sample_block(jiffies, jps, channels, bits, 1);
}

int ohp_jiffies_round_down(uint32_t* jiffies, uint32_t sample_rate)
{                                                                     /* Msg.cpp:491-495 */
    const int jps = ohp_jiffies_per_sample(sample_rate);
    if (jps < 0) return jps;
    *jiffies -= *jiffies % (uint32_t)jps;
    return OHP_OK;
}

int ohp_jiffies_round_up(uint32_t* jiffies, uint32_t sample_rate)
{                                                                     /* Msg.cpp:497-502 */
    const int jps = ohp_jiffies_per_sample(sample_rate);
    if (jps < 0) return jps;
    *jiffies += (uint32_t)jps - 1;
    *jiffies -= *jiffies % (uint32_t)jps;
    return OHP_OK;
}

void ohp_jiffies_round_down_nonzero_sample_block(uint32_t* jiffies, uint32_t block)
{                                                                     /* Msg.cpp:504-514 */
    uint32_t j = *jiffies;
    j -= j % block;
    if (j == 0) {
        j = *jiffies;
        j += block - 1;
        j -= j % block;
    }
    *jiffies = j;
}

int ohp_jiffies_to_songcast_time(uint32_t jiffies, uint32_t sample_rate, uint32_t* out)
{                                                                     /* Msg.cpp:516-558 */
    uint32_t ticks;
    switch (sample_rate) {
    case 7350: case 11025: case 14700: case 22050: case 29400: case 44100: case 88200: case 176400: case 352800:
        ticks = 44100u * 256u; break;
    case 8000: case 12000: case 16000: case 24000: case 32000: case 48000: case 96000: case 192000: case 384000:
        ticks = 48000u * 256u; break;
    default:
        return OHP_ERR_SAMPLE_RATE;
    }
    *out = (uint32_t)(((uint64_t)jiffies * ticks) / OHP_JIFFIES_PER_SEC);
    return OHP_OK;
}

/* ------------------------------------------------------------------------------------------
 * a3  Ramp (Msg.cpp:569-807)
 * ---------------------------------------------------------------------------------------- */
void ohp_ramp_reset(ohp_ramp* r)
{                                                                     /* Msg.cpp:582-588 */
    r->start = OHP_RAMP_MAX;
    r->end = OHP_RAMP_MAX;
    r->direction = OHP_RAMP_NONE;
    r->enabled = 0;
}

int ohp_ramp_validate(const ohp_ramp* r)
{                                                                     /* Msg.cpp:745-782 */
    if (r->start > OHP_RAMP_MAX) return 0;
    if (r->end > OHP_RAMP_MAX) return 0;
    switch (r->direction) {
    case OHP_RAMP_NONE: if (r->start != r->end) return 0; break;
    case OHP_RAMP_UP:   if (r->start >= r->end) return 0; break;
    case OHP_RAMP_DOWN: if (r->start <= r->end) return 0; break;
    case OHP_RAMP_MUTE: if (r->start != r->end) return 0; if (r->start != OHP_RAMP_MIN) return 0; break;
    default: return 0; /* reference ASSERTS() */
    }
    return 1;
}

static void select_lower_ramp_points(ohp_ramp* r, uint32_t req_start, uint32_t req_end)
{                                                                     /* Msg.cpp:721-734 */
    if (req_start < r->start) r->start = req_start;
    if (req_end < r->end) r->end = req_end;
    if (r->start == r->end) r->direction = OHP_RAMP_NONE;
    else if (r->start > r->end) r->direction = OHP_RAMP_DOWN;
    else r->direction = OHP_RAMP_UP;
}

int ohp_ramp_set(ohp_ramp* r, uint32_t aStart, uint32_t aFragmentSize, uint32_t aRemainingDuration,
                 uint32_t aDirection, ohp_ramp* aSplit, uint32_t* aSplitPos)
{                                                                     /* Msg.cpp:590-712 */
    if (!(aRemainingDuration >= aFragmentSize)) return OHP_ERR_ASSERT;   /* :598 */
    if (aDirection == OHP_RAMP_NONE) return OHP_ERR_ASSERT;              /* :599 */
    r->enabled = 1;
    ohp_ramp_reset(aSplit);
    *aSplitPos = 0xffffffffu;
    const uint32_t rampRemaining = (aDirection == OHP_RAMP_DOWN ? aStart : OHP_RAMP_MAX - aStart);
    /* round up (:605) */
    uint32_t rampDelta = (uint32_t)((((uint64_t)rampRemaining * (uint64_t)aFragmentSize) + aRemainingDuration - 1) / aRemainingDuration);
    uint32_t rampEnd;
    if (aDirection == OHP_RAMP_DOWN) {
        if (rampDelta > aStart) {
            if (!(rampDelta - aStart <= aFragmentSize - 1)) return OHP_ERR_ASSERT;   /* :611 */
            rampEnd = 0;
        } else {
            rampEnd = aStart - rampDelta;
        }
    } else { /* EUp (EMute falls here too, as in the reference) */
        if (aStart + rampDelta > OHP_RAMP_MAX) {
            if (!(aStart + rampDelta - OHP_RAMP_MAX <= aFragmentSize - 1)) return OHP_ERR_ASSERT; /* :620 */
            rampEnd = OHP_RAMP_MAX;
        } else {
            rampEnd = aStart + rampDelta;
        }
    }
    if (r->direction == OHP_RAMP_NONE) {                               /* :627-632 */
        r->direction = aDirection;
        r->start = aStart;
        r->end = rampEnd;
    } else if (r->direction == aDirection) {                           /* :633-636 */
        select_lower_ramp_points(r, aStart, rampEnd);
    } else {                                                           /* :637-701 */
        int64_t y1, y2, y3, y4;
        if (r->start < aStart) { y1 = r->start; y2 = r->end; y3 = aStart; y4 = rampEnd; }
        else                   { y1 = aStart; y2 = rampEnd; y3 = r->start; y4 = r->end; }
        if ((y2 - y1) == (y4 - y3)) {
            select_lower_ramp_points(r, aStart, rampEnd);
        } else {
            const int64_t intersectX = ((int64_t)aFragmentSize * (y3 - y1)) / ((y2 - y1) - (y4 - y3));
            const int64_t intersectY = (((y2 - y1) * (y3 - y1)) / ((y2 - y1) - (y4 - y3))) + y1;
            if (intersectX <= 0 || (uint32_t)intersectX >= aFragmentSize) {
                select_lower_ramp_points(r, aStart, rampEnd);
            } else {
                *aSplitPos = (uint32_t)intersectX;
                aSplit->start = (uint32_t)intersectY;
                aSplit->end = (r->end < rampEnd ? r->end : rampEnd);
                aSplit->direction = (aSplit->start == aSplit->end ? OHP_RAMP_NONE : OHP_RAMP_DOWN);
                aSplit->enabled = 1;
                const uint32_t start = (r->start < aStart ? r->start : aStart);
                const uint32_t end = (uint32_t)intersectY;
                r->direction = (start == end ? OHP_RAMP_NONE : OHP_RAMP_UP);
                r->start = start;
                r->end = end;
            }
        }
    }
    if (!ohp_ramp_validate(r)) return OHP_ERR_ASSERT;                  /* :703-710 */
    return aSplit->enabled ? 1 : 0;
}

void ohp_ramp_set_muted(ohp_ramp* r)
{                                                                     /* Msg.cpp:714-719 */
    r->start = r->end = OHP_RAMP_MIN;
    r->direction = OHP_RAMP_MUTE;
    r->enabled = 1;
}

int ohp_ramp_split(ohp_ramp* r, uint32_t aNewSize, uint32_t aCurrentSize, ohp_ramp* remaining)
{                                                                     /* Msg.cpp:784-807 */
    ohp_ramp_reset(remaining);
    remaining->end = r->end;
    remaining->direction = r->direction;
    remaining->enabled = 1;
    if (r->direction == OHP_RAMP_UP) {
        const uint32_t ramp = (uint32_t)(((uint64_t)(r->end - r->start) * (uint64_t)aNewSize) / aCurrentSize);
        r->end = r->start + ramp;
    } else {
        const uint32_t ramp = (uint32_t)(((uint64_t)(r->start - r->end) * (uint64_t)aNewSize) / aCurrentSize);
        r->end = r->start - ramp;
    }
    if (r->start == r->end) r->direction = OHP_RAMP_NONE;
    remaining->start = r->end;
    if (!ohp_ramp_validate(r)) return OHP_ERR_ASSERT;
    if (!ohp_ramp_validate(remaining)) return OHP_ERR_ASSERT;
    return OHP_OK;
}

uint32_t ohp_ramp_median_multiplier(const ohp_ramp* r)
{                                                                     /* Msg.cpp:901-920 */
    uint32_t med;
    switch (r->direction) {
    case OHP_RAMP_UP:   med = r->start + ((r->end - r->start) / 2); break;
    case OHP_RAMP_DOWN: med = r->start - ((r->start - r->end) / 2); break;
    case OHP_RAMP_MUTE: return 0;
    default:            med = r->start; break;
    }
    const uint32_t idx = (OHP_RAMP_MAX - OHP_RAMP_MIN - med + (1u << 4)) >> 5;
    return idx < OHP_RAMP_TABLE_COUNT ? ohp_ramp_table()[idx] : 0; /* idx==512 reads past the array in the reference */
}

/* ------------------------------------------------------------------------------------------
 * a4/a5  MsgAudio metadata (Msg.cpp:1949-2074, 2155-2168, 2234-2276, 2466-2491, 2547-2560)
 * ---------------------------------------------------------------------------------------- */
int ohp_msg_audio_init_pcm(ohp_msg_audio* m, uint32_t data_bytes, uint32_t channels, uint32_t sample_rate, uint32_t bit_depth)
{
    const int jps = ohp_jiffies_per_sample(sample_rate);
    if (jps < 0) return jps;
    const uint32_t byte_depth = bit_depth / 8;
    if (byte_depth == 0 || data_bytes % byte_depth != 0) return OHP_ERR_ASSERT;     /* Msg.cpp:2270 */
    const uint32_t num_subsamples = data_bytes / byte_depth;
    if (channels == 0 || num_subsamples % channels != 0) return OHP_ERR_ASSERT;     /* Msg.cpp:2164 */
    memset(m, 0, sizeof(*m));
    ohp_ramp_reset(&m->ramp);
    m->sample_rate = sample_rate;
    m->bit_depth = bit_depth;
    m->channels = channels;
    m->size_jiffies = (num_subsamples / channels) * (uint32_t)jps;                  /* Msg.cpp:2165 */
    if (m->size_jiffies == 0) return OHP_ERR_ASSERT;                                /* Msg.cpp:2166 */
    m->offset_jiffies = 0;
    m->attenuation = OHP_UNITY_ATTENUATION;
    m->is_silence = 0;
    return OHP_OK;
}

int ohp_msg_audio_init_silence(ohp_msg_audio* m, uint32_t* jiffies, uint32_t sample_rate, uint32_t bit_depth, uint32_t channels)
{
    const int jps = ohp_jiffies_per_sample(sample_rate);
    if (jps < 0) return jps;
    memset(m, 0, sizeof(*m));
    ohp_ramp_reset(&m->ramp);
    m->sample_rate = sample_rate;
    m->bit_depth = bit_depth;
    m->channels = channels;
    ohp_jiffies_round_down_nonzero_sample_block(jiffies, (uint32_t)jps);            /* Msg.cpp:2556 */
    m->size_jiffies = *jiffies;
    m->offset_jiffies = 0;
    m->attenuation = OHP_UNITY_ATTENUATION;
    m->is_silence = 1;
    return OHP_OK;
}

int ohp_msg_audio_split(ohp_msg_audio* m, uint32_t jiffies, ohp_msg_audio* remaining)
{                                                                     /* Msg.cpp:1949-1969 */
    if (!(jiffies > 0)) return OHP_ERR_ASSERT;
    if (!(jiffies < m->size_jiffies)) return OHP_ERR_ASSERT;
    *remaining = *m;
    remaining->offset_jiffies = m->offset_jiffies + jiffies;
    remaining->size_jiffies = m->size_jiffies - jiffies;
    if (m->ramp.enabled) {
        const int err = ohp_ramp_split(&m->ramp, jiffies, m->size_jiffies, &remaining->ramp);
        if (err < 0) return err;
    } else {
        ohp_ramp_reset(&remaining->ramp);
    }
    m->size_jiffies = jiffies;
    if (m->is_silence) {                                              /* MsgSilence::SplitCompleted, Msg.cpp:2520-2545 */
        const uint32_t block = (uint32_t)ohp_jiffies_per_sample(m->sample_rate);
        const uint32_t rem = m->size_jiffies % block;
        m->size_jiffies -= rem;
        remaining->size_jiffies += rem;
    }
    return OHP_OK;
}

int ohp_msg_audio_set_ramp(ohp_msg_audio* m, uint32_t aStart, uint32_t* aRemainingDuration, uint32_t aDirection,
                           ohp_msg_audio* aSplit, int* has_split, uint32_t* ramp_end_out)
{                                                                     /* Msg.cpp:1989-2046 */
    const uint32_t remainingDuration = *aRemainingDuration;
    ohp_ramp split;
    uint32_t splitPos;
    *has_split = 0;
    if (!(aDirection == OHP_RAMP_UP || aDirection == OHP_RAMP_DOWN)) return OHP_ERR_ASSERT;
    if (m->ramp.enabled && m->ramp.direction == OHP_RAMP_MUTE) {
        if (aDirection == OHP_RAMP_DOWN) *aRemainingDuration = 0;
        *ramp_end_out = m->ramp.end;
        return OHP_OK;
    }
    const int rs = ohp_ramp_set(&m->ramp, aStart, m->size_jiffies, remainingDuration, aDirection, &split, &splitPos);
    if (rs < 0) return rs;
    if (rs == 1) {
        if (splitPos == 0) {
            m->ramp = split;
        } else if (splitPos != m->size_jiffies) {
            const ohp_ramp ramp = m->ramp;
            const int err = ohp_msg_audio_split(m, splitPos, aSplit);
            if (err < 0) return err;
            *has_split = 1;
            m->ramp = ramp;
            aSplit->ramp = split;
        }
    }
    *aRemainingDuration -= m->size_jiffies;
    if (*has_split && aSplit->ramp.direction != aDirection && aDirection == OHP_RAMP_UP) {
        *aRemainingDuration += aSplit->size_jiffies;
    }
    if (aDirection == OHP_RAMP_DOWN && m->ramp.end == OHP_RAMP_MIN) *aRemainingDuration = 0;
    else if (aDirection == OHP_RAMP_UP && m->ramp.end == OHP_RAMP_MAX) *aRemainingDuration = 0;
    *ramp_end_out = m->ramp.end;
    return OHP_OK;
}

int ohp_create_playable(const ohp_msg_audio* m, ohp_playable* p)
{
    const int jps_i = ohp_jiffies_per_sample(m->sample_rate);
    if (jps_i < 0) return jps_i;
    const uint32_t jps = (uint32_t)jps_i;
    memset(p, 0, sizeof(*p));
    p->jiffies = m->size_jiffies;
    p->sample_rate = m->sample_rate;
    p->bit_depth = m->bit_depth;
    p->channels = m->channels;
    p->attenuation = m->attenuation;
    if (m->is_silence) {                                              /* MsgSilence::CreatePlayable, Msg.cpp:2466-2478 */
        uint32_t size_total = m->size_jiffies;
        p->size_bytes = ohp_jiffies_to_bytes(&size_total, jps, m->channels, m->bit_depth);
        p->offset_bytes = 0;
        p->is_silence = 1;
        p->ramp = m->ramp;
        return OHP_OK;
    }
    uint32_t offsetJiffies = m->offset_jiffies;                       /* Msg.cpp:2236-2240 */
    const uint32_t offsetBytes = ohp_jiffies_to_bytes(&offsetJiffies, jps, m->channels, m->bit_depth);
    uint32_t sizeJiffies = m->size_jiffies + (m->offset_jiffies - offsetJiffies);
    const uint32_t sizeBytes = ohp_jiffies_to_bytes(&sizeJiffies, jps, m->channels, m->bit_depth);
    p->offset_bytes = offsetBytes;
    p->size_bytes = sizeBytes;
    if (m->ramp.direction != OHP_RAMP_MUTE) {                          /* Msg.cpp:2245-2251 */
        p->is_silence = 0;
        p->ramp = m->ramp;
    } else {                                                           /* Msg.cpp:2252-2258 */
        p->is_silence = 1;
        p->offset_bytes = 0;
        ohp_ramp_reset(&p->ramp);
    }
    return OHP_OK;
}

int ohp_playable_split(ohp_playable* p, uint32_t aBytes, ohp_playable* remaining, int* has_remaining)
{                                                                     /* Msg.cpp:2591-2624 */
    *has_remaining = 0;
    if (!(aBytes <= p->size_bytes)) return OHP_ERR_ASSERT;
    if (aBytes == 0) return OHP_ERR_ASSERT;
    if (aBytes == p->size_bytes) return OHP_OK;
    const int jps = ohp_jiffies_per_sample(p->sample_rate);
    if (jps < 0) return jps;
    const uint32_t numSamples = aBytes / ((p->bit_depth / 8) * p->channels);
    const uint32_t splitJiffies = numSamples * (uint32_t)jps;
    *remaining = *p;
    remaining->offset_bytes = p->offset_bytes + aBytes;
    remaining->size_bytes = p->size_bytes - aBytes;
    remaining->jiffies = p->jiffies - splitJiffies;
    if (p->ramp.enabled) {
        const int err = ohp_ramp_split(&p->ramp, aBytes, p->size_bytes, &remaining->ramp);
        if (err < 0) return err;
    } else {
        ohp_ramp_reset(&remaining->ramp);
    }
    p->size_bytes = aBytes;
    p->jiffies = splitJiffies;
    *has_remaining = 1;
    return OHP_OK;
}

/* ------------------------------------------------------------------------------------------
 * a6  MsgPlayablePcm::ApplyAttenuation (Msg.cpp:2736-2751)
 * The multiply is TInt * TUint -> unsigned, the divide is unsigned; do not "fix" it.
 * ---------------------------------------------------------------------------------------- */
int ohp_apply_attenuation(uint8_t* data, uint32_t bytes, uint32_t bit_depth, uint32_t attenuation)
{
    if (attenuation == OHP_UNITY_ATTENUATION) return OHP_OK;
    if (bit_depth != 16) return OHP_ERR_ASSERT;                       /* Msg.cpp:2741 */
    const uint32_t samples = bytes / 2;
    uint8_t* ptr = data;
    for (uint32_t i = 0; i < samples; i++) {
        int16_t sample = (int16_t)(uint16_t)(((uint32_t)ptr[0] << 8) + ptr[1]);
        const uint32_t prod = (uint32_t)(int32_t)sample * attenuation;
        const int16_t att = (int16_t)(uint16_t)(prod / OHP_UNITY_ATTENUATION);
        *ptr++ = (uint8_t)((uint16_t)att >> 8);
        *ptr++ = (uint8_t)att;
    }
    return OHP_OK;
}

/* ------------------------------------------------------------------------------------------
 * a7  RampApplicator::Start / GetNextSample (Msg.cpp:816-899)
 * ---------------------------------------------------------------------------------------- */
int ohp_ramp_apply(const uint8_t* src, uint32_t bytes, uint32_t bit_depth, uint32_t channels,
                   uint32_t ramp_start, uint32_t ramp_end, uint8_t* dst)
{
    if (!(bit_depth == 8 || bit_depth == 16 || bit_depth == 24 || bit_depth == 32)) return OHP_ERR_ASSERT;
    if (channels == 0) return OHP_ERR_ASSERT;
    const uint16_t* table = ohp_ramp_table();
    const uint32_t byte_depth = bit_depth / 8;
    const int32_t iNumSamples = (int32_t)(bytes / (byte_depth * channels));        /* :826 */
    const int32_t iTotalRamp = (int32_t)(ramp_start - ramp_end);                   /* :827 */
    const uint8_t* p = src;
    for (int32_t iLoopCount = 0; iLoopCount < iNumSamples; iLoopCount++) {
        const uint16_t ramp = (iNumSamples == 1) ? (uint16_t)ramp_start
            : (uint16_t)(ramp_start - (uint32_t)((iLoopCount * iTotalRamp) / (iNumSamples - 1)));  /* :835 */
        uint32_t rampIndex = (OHP_RAMP_MAX - (uint32_t)ramp + (1u << 4)) >> 5;     /* :837 */
        if (rampIndex > OHP_RAMP_TABLE_COUNT - 1) rampIndex = OHP_RAMP_TABLE_COUNT - 1;
        const uint16_t rampMult = table[rampIndex];
        for (uint32_t i = 0; i < channels; i++) {
            int16_t subsample16;
            switch (bit_depth) {                                                   /* :840-862 */
            case 8:  subsample16 = (int16_t)(uint16_t)((uint32_t)p[0] << 8); p += 1; break;
            case 16: subsample16 = (int16_t)(uint16_t)(((uint32_t)p[0] << 8) + p[1]); p += 2; break;
            case 24: subsample16 = (int16_t)(uint16_t)(((uint32_t)p[0] << 8) + p[1]); p += 3; break;
            default: subsample16 = (int16_t)(uint16_t)(((uint32_t)p[0] << 8) + p[1]); p += 4; break;
            }
            const int32_t ramped = ((int32_t)subsample16 * (int32_t)rampMult) >> 15;  /* :865 (arithmetic shift) */
            switch (bit_depth) {                                                   /* :868-895 */
            case 8:  *dst++ = (uint8_t)(ramped >> 8); break;
            case 16: *dst++ = (uint8_t)(ramped >> 8); *dst++ = (uint8_t)ramped; break;
            case 24: *dst++ = (uint8_t)(ramped >> 8); *dst++ = (uint8_t)ramped; *dst++ = 0; break;
            default:
                *dst++ = (uint8_t)(ramped >> 8); *dst++ = (uint8_t)ramped; *dst++ = 0;
                *dst++ = (channels == 6) ? (uint8_t)(((uint8_t)i) << 4) : 0;      /* :885-890 */
                break;
            }
        }
    }
    return OHP_OK;
}

/* ------------------------------------------------------------------------------------------
 * a8/a9  MsgPlayable::Read, MsgPlayablePcm::ReadBlock, MsgPlayableSilence::ReadBlock
 *        (Msg.cpp:2646-2653, 2753-2786, 2874-2893)
 * ---------------------------------------------------------------------------------------- */
static const uint8_t kSilence6chHead[32] = {                          /* Msg.cpp:2877 */
    0, 0, 0, 0x00, 0, 0, 0, 0x10, 0, 0, 0, 0x20, 0, 0, 0, 0x30,
    0, 0, 0, 0x40, 0, 0, 0, 0x50, 0, 0, 0, 0x60, 0, 0, 0, 0x70 };

int ohp_playable_read(const ohp_playable* p, uint8_t* audio, uint8_t* out, uint32_t out_capacity,
                      uint32_t* frag_sizes, uint32_t max_frags, uint32_t* n_frags, uint32_t* out_bytes)
{
    uint32_t nf = 0, ob = 0;
    *n_frags = 0;
    *out_bytes = 0;
    if (p->size_bytes == 0) return OHP_OK;                            /* Msg.cpp:2649 */
    if (p->size_bytes > out_capacity) return OHP_ERR_ASSERT;
    const uint32_t subsampleBytes = p->bit_depth / 8;
    if (p->is_silence) {                                              /* Msg.cpp:2874-2893 */
        uint32_t remaining = p->size_bytes;
        const uint32_t maxBytes = OHP_MAX_BYTES - (OHP_MAX_BYTES % (p->channels * subsampleBytes));
        do {
            const uint32_t bytes = (remaining > maxBytes ? maxBytes : remaining);
            memset(out + ob, 0, bytes);
            if (p->channels == 6) memcpy(out + ob, kSilence6chHead, bytes < 32 ? bytes : 32);
            if (frag_sizes && nf < max_frags) frag_sizes[nf] = bytes;
            nf++;
            ob += bytes;
            remaining -= bytes;
        } while (remaining > 0);
        *n_frags = nf;
        *out_bytes = ob;
        return OHP_OK;
    }
    uint8_t* buf = audio + p->offset_bytes;                           /* Msg.cpp:2755 */
    const int err = ohp_apply_attenuation(buf, p->size_bytes, p->bit_depth, p->attenuation);
    if (err < 0) return err;
    if (p->ramp.enabled) {                                            /* Msg.cpp:2761-2780 */
        const int e2 = ohp_ramp_apply(buf, p->size_bytes, p->bit_depth, p->channels, p->ramp.start, p->ramp.end, out);
        if (e2 < 0) return e2;
        const uint32_t bytesPerSample = subsampleBytes * p->channels;
        const uint32_t samplesPerFragment = 256 / bytesPerSample;
        const uint32_t numSamples = p->size_bytes / bytesPerSample;
        uint32_t done = 0;
        while (done < numSamples) {
            const uint32_t n = (numSamples - done > samplesPerFragment ? samplesPerFragment : numSamples - done);
            if (frag_sizes && nf < max_frags) frag_sizes[nf] = n * bytesPerSample;
            nf++;
            done += n;
        }
        ob = numSamples * bytesPerSample;
    } else {                                                          /* Msg.cpp:2782-2784 */
        memcpy(out, buf, p->size_bytes);
        if (frag_sizes && max_frags > 0) frag_sizes[0] = p->size_bytes;
        nf = 1;
        ob = p->size_bytes;
    }
    *n_frags = nf;
    *out_bytes = ob;
    return OHP_OK;
}

/* ------------------------------------------------------------------------------------------
 * a11  FlywheelInput::DoProcessFragment + AppendSubsample8/16/24/32 (StarvationRamper.cpp:117-186)
 * ---------------------------------------------------------------------------------------- */
int ohp_flywheel_unpack(const uint8_t* data, uint32_t bytes, uint32_t channels, uint32_t subsample_bytes,
                        uint8_t* planes, uint32_t plane_stride_bytes, uint32_t* plane_pos)
{
    if (subsample_bytes < 1 || subsample_bytes > 4) return OHP_ERR_ASSERT;          /* :178-180 */
    const uint8_t* src = data;
    const uint32_t numSubsamples = bytes / subsample_bytes;
    const uint32_t numSamples = numSubsamples / channels;
    for (uint32_t i = 0; i < numSamples; i++) {
        for (uint32_t j = 0; j < channels; j++) {
            uint8_t* d = planes + (size_t)j * plane_stride_bytes + plane_pos[j];
            uint32_t b = 0;
            for (; b < subsample_bytes; b++) d[b] = *src++;
            for (; b < 4; b++) d[b] = 0;
            plane_pos[j] += 4;
        }
    }
    return OHP_OK;
}

/* ------------------------------------------------------------------------------------------
 * a12  RampGenerator::ProcessFragment (StarvationRamper.cpp:281-327)
 * ---------------------------------------------------------------------------------------- */
int ohp_rampgen_pack(const uint8_t* data, uint32_t bytes, uint32_t bit_depth, uint8_t* dst, uint32_t* dst_bytes)
{
    const uint32_t subsamples = bytes / 4;
    const uint8_t* src = data;
    uint8_t* d = dst;
    switch (bit_depth) {
    case 8:  for (uint32_t i = 0; i < subsamples; i++) { *d++ = src[0]; src += 4; } break;
    case 16: for (uint32_t i = 0; i < subsamples; i++) { *d++ = src[0]; *d++ = src[1]; src += 4; } break;
    case 24: for (uint32_t i = 0; i < subsamples; i++) { *d++ = src[0]; *d++ = src[1]; *d++ = src[2]; src += 4; } break;
    case 32: for (uint32_t i = 0; i < subsamples; i++) { *d++ = src[0]; *d++ = src[1]; *d++ = src[2]; *d++ = 0; src += 4; } break;
    default: return OHP_ERR_ASSERT;
    }
    *dst_bytes = subsamples * (bit_depth / 8);
    return OHP_OK;
}

/* ------------------------------------------------------------------------------------------
 * a13  Sender::FirstChannelToSend / DoProcessFragment (Av/Songcast/Sender.cpp:351-377)
 * The reference always copies two subsamples per frame and advances dst by the output channel
 * count; for mono input the second copy is overwritten by the next frame.  The final frame's
 * stray copy lands outside the counted bytes and is not reproduced here.
 * ---------------------------------------------------------------------------------------- */
int ohp_sender_pack(const uint8_t* data, uint32_t bytes, uint32_t channels, uint32_t bps, uint8_t* dst, uint32_t* dst_bytes)
{
    if (channels == 0 || bps == 0) return OHP_ERR_ASSERT;
    const uint32_t first = (channels < 10) ? 0 : 8;
    const uint8_t* src = data + bps * first;
    const uint32_t stride = bps * channels;
    const uint32_t numSamples = bytes / stride;
    const uint32_t dstBps = bps < 3 ? bps : 3;
    const uint32_t outCh = channels < 2 ? channels : 2;
    uint8_t* d = dst;
    for (uint32_t i = 0; i < numSamples; i++) {
        memcpy(d, src, dstBps);
        if (outCh == 2) memcpy(d + dstBps, src + bps, dstBps);
        src += stride;
        d += outCh * dstBps;
    }
    *dst_bytes = numSamples * outCh * dstBps;
    return OHP_OK;
}

/* ------------------------------------------------------------------------------------------
 * a14  CodecFlac::CallbackWrite packer (Codec/Flac.cpp:379-417)
 * ---------------------------------------------------------------------------------------- */
int ohp_flac_pack(const int32_t* const* planes, uint32_t channels, uint32_t first, uint32_t samples,
                  uint32_t bit_depth, uint8_t* dst, uint32_t* dst_bytes)
{
    uint8_t* p = dst;
    if (!(bit_depth == 8 || bit_depth == 16 || bit_depth == 24)) return OHP_ERR_UNSUPPORTED;   /* :404-407 */
    for (uint32_t i = first; i < first + samples; i++) {
        for (uint32_t j = 0; j < channels; j++) {
            const uint32_t subsample = (uint32_t)planes[j][i];
            switch (bit_depth) {
            case 8:  *p++ = (uint8_t)subsample; break;
            case 16: *p++ = (uint8_t)(subsample >> 8); *p++ = (uint8_t)subsample; break;
            default: *p++ = (uint8_t)(subsample >> 16); *p++ = (uint8_t)(subsample >> 8); *p++ = (uint8_t)subsample; break;
            }
        }
    }
    *dst_bytes = samples * channels * (bit_depth / 8);
    return OHP_OK;
}

/* ------------------------------------------------------------------------------------------
 * format helpers (compositions of a1, a11, a12)
 * ---------------------------------------------------------------------------------------- */
int ohp_unpack_s24(const uint8_t* src, uint32_t subsamples, uint32_t bit_depth, int endian, int32_t* dst)
{
    const uint32_t bd = bit_depth / 8;
    if (bd < 1 || bd > 4 || (bit_depth & 7)) return OHP_ERR_ASSERT;
    for (uint32_t i = 0; i < subsamples; i++) {
        uint8_t be[4] = { 0, 0, 0, 0 };
        for (uint32_t b = 0; b < bd; b++)                             /* a1 then a11: left-justified BE */
            be[b] = (endian == OHP_ENDIAN_BIG || bd == 1) ? src[i * bd + b] : src[i * bd + (bd - 1 - b)];
        const uint32_t w = ((uint32_t)be[0] << 24) | ((uint32_t)be[1] << 16) | ((uint32_t)be[2] << 8) | be[3];
        dst[i] = ((int32_t)w) >> 8;                                    /* top 24 bits, sign-extended */
    }
    return OHP_OK;
}

int ohp_pack_from_s24(const int32_t* src, uint32_t subsamples, uint32_t bit_depth, int endian, uint8_t* dst)
{
    const uint32_t bd = bit_depth / 8;
    if (bd < 1 || bd > 4 || (bit_depth & 7)) return OHP_ERR_ASSERT;
    for (uint32_t i = 0; i < subsamples; i++) {
        const uint32_t w = ((uint32_t)src[i]) << 8;                   /* left-justify to 32 (a11) */
        uint8_t be[4] = { (uint8_t)(w >> 24), (uint8_t)(w >> 16), (uint8_t)(w >> 8), 0 };  /* a12: "32" keeps 3 bytes + 0 */
        for (uint32_t b = 0; b < bd; b++)
            dst[i * bd + b] = (endian == OHP_ENDIAN_BIG) ? be[b] : be[bd - 1 - b];
    }
    return OHP_OK;
}

int ohp_convert_format(const uint8_t* src, uint32_t subsamples, uint32_t src_bits, int src_endian,
                       uint32_t dst_bits, int dst_endian, int zero_lsb32, uint8_t* dst)
{
    const uint32_t sb = src_bits / 8, db = dst_bits / 8;
    if (sb < 1 || sb > 4 || db < 1 || db > 4) return OHP_ERR_ASSERT;
    for (uint32_t i = 0; i < subsamples; i++) {
        uint8_t be[4] = { 0, 0, 0, 0 };
        for (uint32_t b = 0; b < sb; b++)                             /* a1 + a11 */
            be[b] = (src_endian == OHP_ENDIAN_BIG || sb == 1) ? src[i * sb + b] : src[i * sb + (sb - 1 - b)];
        if (db == 4 && zero_lsb32) be[3] = 0;                         /* a12 case 32 */
        for (uint32_t b = 0; b < db; b++)
            dst[i * db + b] = (dst_endian == OHP_ENDIAN_BIG) ? be[b] : be[db - 1 - b];
    }
    return OHP_OK;
}

/* ------------------------------------------------------------------------------------------
 * a-R  polyphase sample-rate converter.  NO REFERENCE -- PARITY UNPINNED (ohp_oracle.h).
 * ---------------------------------------------------------------------------------------- */
static uint32_t gcd_u32(uint32_t a, uint32_t b) { while (b) { uint32_t t = a % b; a = b; b = t; } return a; }

static double bessel_i0(double x)
{
    double sum = 1.0, term = 1.0;
    const double q = x * x * 0.25;
    for (int k = 1; k < 500; k++) {
        term *= q / ((double)k * (double)k);
        sum += term;
        if (term < sum * 1e-20) break;
    }
    return sum;
}

int ohp_src_design(ohp_src* s, uint32_t rate_in, uint32_t rate_out, uint32_t T, double beta, double f_pass)
{
    memset(s, 0, sizeof(*s));
    if (rate_in == 0 || rate_out == 0 || T == 0) return OHP_ERR_ASSERT;
    const uint32_t g = gcd_u32(rate_in, rate_out);
    const uint32_t L = rate_out / g, M = rate_in / g;
    /* an integer decimator (L = 1) gets an odd length, T - 1, centred on a tap, stored with coef[T - 1] = 0 (for 2:1 with the
     * cutoff at a quarter of the input rate that is a half-band filter: every second coefficient rounds to exactly zero) */
    const uint32_t N = (L == 1 && T > 1) ? T - 1 : L * T;
    s->L = L; s->M = M; s->T = T; s->beta = beta; s->rate_in = rate_in; s->rate_out = rate_out;
    s->f_pass = f_pass;
    s->f_stop = (double)rate_out - f_pass;
    if (s->f_stop > (double)rate_in - f_pass && rate_out > 2 * rate_in) s->f_stop = (double)rate_in - f_pass;
    const double fs_up = (double)L * (double)rate_in;
    const double fc = 0.5 * (s->f_pass + s->f_stop);
    const double wc = 2.0 * fc / fs_up;                 /* cutoff as a fraction of the upsampled Nyquist */
    const double centre = 0.5 * (double)(N - 1);
    const double i0b = bessel_i0(beta);
    double* h = (double*)malloc(sizeof(double) * N);
    s->coef_q28 = (int32_t*)malloc(sizeof(int32_t) * L * T);
    s->coef_f64 = (double*)malloc(sizeof(double) * L * T);
    if (!h || !s->coef_q28 || !s->coef_f64) { free(h); ohp_src_free(s); return OHP_ERR_ASSERT; }
    double sum = 0.0;
    for (uint32_t n = 0; n < N; n++) {
        const double d = (double)n - centre;
        const double x = wc * d;
        const double sinc = (fabs(x) < 1e-12) ? 1.0 : sin(M_PI * x) / (M_PI * x);
        const double r = (centre > 0.0) ? d / centre : 0.0;
        const double arg = 1.0 - r * r;
        const double w = bessel_i0(beta * sqrt(arg > 0.0 ? arg : 0.0)) / i0b;
        h[n] = wc * sinc * w;
        sum += h[n];
    }
    const double scale = (double)L / sum;               /* overall DC gain L (unity per phase on average) */
    s->sum_abs_max = 0;
    for (uint32_t p = 0; p < L; p++) {
        int64_t sabs = 0;
        for (uint32_t k = 0; k < T; k++) {
            const double v = (p + k * L < N) ? h[p + k * L] * scale : 0.0;
            const int32_t q = (int32_t)floor(v * 268435456.0 + 0.5);
            s->coef_f64[p * T + k] = v;
            s->coef_q28[p * T + k] = q;
            sabs += (q < 0 ? -(int64_t)q : (int64_t)q);
        }
        if (sabs > s->sum_abs_max) s->sum_abs_max = sabs;
    }
    free(h);
    if (s->sum_abs_max >= ((int64_t)1 << 30)) { ohp_src_free(s); return OHP_ERR_ASSERT; } /* keeps |acc| < 2^53 */
    return OHP_OK;
}

void ohp_src_free(ohp_src* s)
{
    free(s->coef_q28); s->coef_q28 = NULL;
    free(s->coef_f64); s->coef_f64 = NULL;
}

uint64_t ohp_src_out_frames(const ohp_src* s, uint64_t in_frames_total)
{
    /* outputs m with n0(m) = floor(m*M/L) <= in_frames_total-1  <=>  m < ceil(in*L/M) */
    if (in_frames_total == 0) return 0;
    return (in_frames_total * s->L + s->M - 1) / s->M;
}

int ohp_src_process_i64(const ohp_src* s, const int32_t* x, int64_t x_first, uint64_t x_frames, uint32_t channels,
                        uint64_t m0, uint32_t n_out, int32_t* y)
{
    const uint32_t L = s->L, M = s->M, T = s->T;
    for (uint32_t o = 0; o < n_out; o++) {
        const uint64_t t = (m0 + o) * (uint64_t)M;
        const int64_t n0 = (int64_t)(t / L);
        const uint32_t p = (uint32_t)(t % L);
        const int32_t* c = s->coef_q28 + (size_t)p * T;
        for (uint32_t ch = 0; ch < channels; ch++) {
            int64_t acc = 0;
            for (uint32_t k = 0; k < T; k++) {
                const int64_t idx = n0 - (int64_t)k;
                if (idx < 0) continue;                               /* before stream start: zeros */
                const int64_t rel = idx - x_first;
                if (rel < 0 || (uint64_t)rel >= x_frames) return OHP_ERR_ASSERT;
                acc += (int64_t)c[k] * (int64_t)x[(size_t)rel * channels + ch];
            }
            int64_t v = (acc + ((int64_t)1 << 27)) >> 28;            /* round half up (floor of acc/2^28 + 0.5) */
            if (v > 8388607) v = 8388607;
            if (v < -8388608) v = -8388608;
            y[(size_t)o * channels + ch] = (int32_t)v;
        }
    }
    return OHP_OK;
}

int ohp_src_process_f64(const ohp_src* s, const int32_t* x, int64_t x_first, uint64_t x_frames, uint32_t channels,
                        uint64_t m0, uint32_t n_out, double* y)
{
    const uint32_t L = s->L, M = s->M, T = s->T;
    for (uint32_t o = 0; o < n_out; o++) {
        const uint64_t t = (m0 + o) * (uint64_t)M;
        const int64_t n0 = (int64_t)(t / L);
        const uint32_t p = (uint32_t)(t % L);
        const double* c = s->coef_f64 + (size_t)p * T;
        for (uint32_t ch = 0; ch < channels; ch++) {
            double acc = 0.0;
            for (uint32_t k = 0; k < T; k++) {
                const int64_t idx = n0 - (int64_t)k;
                if (idx < 0) continue;
                const int64_t rel = idx - x_first;
                if (rel < 0 || (uint64_t)rel >= x_frames) return OHP_ERR_ASSERT;
                acc += c[k] * (double)x[(size_t)rel * channels + ch];
            }
            y[(size_t)o * channels + ch] = acc;
        }
    }
    return OHP_OK;
}
