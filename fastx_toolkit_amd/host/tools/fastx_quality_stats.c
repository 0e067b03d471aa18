/* fastx_quality_stats -- command line and both report formats of the reference tool (src/fastx_quality_stats/fastx_quality_stats.c:296-414).
 *
 * FASTQ input is reduced on the GPU to hist[column][A,C,G,T,N][quality byte] (fxg_run_quality_stats); the report is a function of
 * that histogram.  FASTA input has bases but no qualities: counts per column and class, weighted by the collapsed-read multiplicity,
 * come from the record API.
 *
 * The numbers are kept the way the reference's arithmetic sees them: one flat image of `int`s, REC ints per (column, class) record --
 *     [0] smallest quality (100 while empty)   [1] largest (-100 while empty)   [2] bases   [3..4] sum of qualities (64 bit)
 *     [5 .. 5 + QUALITY_VALUES_RANGE)  bases per quality value, bin = quality - MIN_QUALITY_VALUE
 * records in (column, class) order, class 0 = all bases.  Two behaviours of the reference fall out of this layout and nothing else
 * (its structs are byte-packed because fastx.h:61 leaves `#pragma pack(1)` on, so a record is exactly these REC ints):
 *   * quality 93 has bin QUALITY_VALUES_RANGE, one past a record's bins: it is counted in field [0] of the NEXT record;
 *   * the k-th smallest quality of a record is found by walking its bins upwards (fastx_quality_stats.c:237-243).  With FASTA input
 *     a record has bases and no qualities, so the walk leaves the record and runs through the fields of the records behind it.
 * cell() below is that image, including the untouched records behind the last column of the input (the reference's array has
 * MAX_SEQ_LINE_LENGTH columns); the walk stops at the end of the image, where the reference would read whatever the linker put next.
 */
#include <err.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../fastx.h"
#include "../fastx_args.h"
#include "../fxh_batch.h"

const char *usage =
    "usage: fastx_quality_stats [-h] [-N] [-i INFILE] [-o OUTFILE]\n"
    "MI355X build of the FASTX-Toolkit quality statistics tool (same flags as FASTX Toolkit 0.0.14).\n\n"
    "   [-h] = This helpful help screen.\n"
    "   [-i INFILE]  = FASTQ input file. default is STDIN.\n"
    "   [-o OUTFILE] = TEXT output file. default is STDOUT.\n"
    "   [-N]         = New output format (with more information per nucleotide/cycle).\n\n"
    "Old format: column count min max sum mean Q1 med Q3 IQR lW rW A_Count C_Count G_Count T_Count N_Count Max_count\n"
    "New format: cycle max_count, then count min max sum mean Q1 med Q3 IQR lW rW for each of ALL A C G T N\n\n";

enum { CLASSES = 6, REC = 5 + QUALITY_VALUES_RANGE, F_LOW = 0, F_HIGH = 1, F_BASES = 2, F_SUM = 3, F_BIN0 = 5 };
static const char *const class_name[CLASSES] = {"ALL", "A", "C", "G", "T", "N"};
static const char *const column_name[11] = {"count", "min", "max", "sum", "mean", "Q1", "med", "Q3", "IQR", "lW", "rW"};

static int *image;                  /* REC ints per record, records_held of them */
static size_t columns_held;         /* columns of the input (records_held = columns_held * CLASSES) */
static const size_t image_records = (size_t)MAX_SEQ_LINE_LENGTH * CLASSES;

static void hold_columns(size_t n)
{
    if (n > MAX_SEQ_LINE_LENGTH) errx(1, "Internal error: sequence too long. Hard-coded max. length is %d", MAX_SEQ_LINE_LENGTH);
    if (n <= columns_held) return;
    image = realloc(image, n * CLASSES * REC * sizeof(int));
    if (!image) err(1, "out of memory");
    for (size_t r = columns_held * CLASSES; r < n * CLASSES; ++r) {
        memset(image + r * REC, 0, REC * sizeof(int));
        image[r * REC + F_LOW] = 100;
        image[r * REC + F_HIGH] = -100;
    }
    columns_held = n;
}

/* int number i of the image; records nobody touched read as they were initialised */
static int cell(size_t i)
{
    if (i < columns_held * CLASSES * REC) return image[i];
    const size_t field = i % REC;
    return field == F_LOW ? 100 : field == F_HIGH ? -100 : 0;
}
static size_t record(size_t column, int cls) { return (column * CLASSES + (size_t)cls) * REC; }
static long long quality_sum(size_t rec) { unsigned long long s; memcpy(&s, image + rec + F_SUM, sizeof s); return (long long)s; }

/* `bases` bases of quality q in one (column, class): the record itself and the all-bases record of the column */
static void tally(size_t column, int cls, int q, long long bases)
{
    const size_t recs[2] = {record(column, 0), record(column, cls)};
    for (int k = 0; k < 2; ++k) {
        int *r = image + recs[k];
        unsigned long long s;
        if (q < r[F_LOW]) r[F_LOW] = q;
        if (q > r[F_HIGH]) r[F_HIGH] = q;
        r[F_BASES] += (int)bases;
        memcpy(&s, r + F_SUM, sizeof s);
        s += (unsigned long long)((long long)q * bases);
        memcpy(r + F_SUM, &s, sizeof s);
        const size_t bin = recs[k] + F_BIN0 + (size_t)(q - MIN_QUALITY_VALUE);      /* q = 93: field [0] of the next record (N2) */
        if (bin < columns_held * CLASSES * REC) image[bin] += (int)bases;
    }
}

/* the device's hist[column][A,C,G,T,N][quality + 33] into the image */
static void take_histogram(const uint64_t *hist, uint32_t cols)
{
    hold_columns(cols < MAX_SEQ_LINE_LENGTH ? cols + 1u : cols);      /* one untouched column behind the input: where a quality of 93 in the last column is counted */
    for (uint32_t c = 0; c < cols; ++c)
        for (int cls = 1; cls < CLASSES; ++cls) {
            const uint64_t *h = hist + ((size_t)c * FXG_QS_CLASSES + (size_t)(cls - 1)) * FXG_QS_BINS;
            for (int byte = 0; byte < FXG_QS_BINS; ++byte)
                if (h[byte]) tally(c, cls, byte - 33, (long long)h[byte]);
        }
}

/* FASTA: bases only (fastx_quality_stats.c:183-190), a record of `>id-count` standing for `count` reads */
static void take_fasta(FASTX *fx)
{
    static const char letters[] = "ACGTN";
    while (fastx_read_next_record(fx)) {
        const size_t L = strlen(fx->nucleotides);
        const int weight = get_reads_count(fx);
        hold_columns(L);
        for (size_t i = 0; i < L; ++i) {
            const char up = (char)(fx->nucleotides[i] & ~0x20);
            const char *hit = up ? strchr(letters, up) : NULL;
            image[record(i, 0) + F_BASES] += weight;
            image[record(i, hit ? (int)(hit - letters) + 1 : 0) + F_BASES] += weight;
        }
    }
}

/* quality of the base of rank n (0-based, ascending) of one record: the reference's walk over the image */
static int ranked_quality(size_t column, int cls, int n)
{
    const size_t rec = record(column, cls);
    const int bases = cell(rec + F_BASES);
    if (n == 0) return cell(rec + F_LOW);
    if (n < 0 || n >= bases) {
        fprintf(stderr, "Internal error at get_nth_value (cycle=%d, nucleotide=%d, n=%d), count_values[%d]=%d\n", (int)column, cls, n, (int)column, bases);
        exit(1);
    }
    const size_t first = rec + F_BIN0, last = image_records * REC - 1;
    size_t at = first;
    while (n > 0) {
        const int here = cell(at);
        if (here > n) break;
        n -= here;
        if (at >= last) break;
        do ++at; while (cell(at) == 0 && at < last);
    }
    return (int)(at - first) + MIN_QUALITY_VALUE;
}

/* the eleven numbers the reports print for one record, in column_name[] order (mean apart: it is the one floating-point field) */
struct summary { long long v[11]; double mean; };
static struct summary summarise(size_t column, int cls)
{
    struct summary s;
    const size_t rec = record(column, cls);
    const int bases = cell(rec + F_BASES), low = cell(rec + F_LOW), high = cell(rec + F_HIGH);
    const int q1 = ranked_quality(column, cls, bases / 4), med = ranked_quality(column, cls, bases / 2), q3 = ranked_quality(column, cls, bases * 3 / 4);
    const int iqr = q3 - q1, reach = iqr * 3 / 2;
    s.v[0] = bases; s.v[1] = low; s.v[2] = high; s.v[3] = quality_sum(rec); s.v[4] = 0;
    s.v[5] = q1; s.v[6] = med; s.v[7] = q3; s.v[8] = iqr;
    s.v[9] = q1 - reach < low ? low : q1 - reach;
    s.v[10] = q3 + reach > high ? high : q3 + reach;
    s.mean = (double)(unsigned long long)s.v[3] / (double)bases;      /* the reference's sum is unsigned: a negative total prints as 2^64 - |sum| here */
    return s;
}
static void print_summary(FILE *out, const struct summary *s)
{
    for (int f = 0; f < 11; ++f) {
        if (f) fputc('\t', out);
        if (f == 4) fprintf(out, "%3.2f", s->mean);
        else fprintf(out, "%lld", s->v[f]);
    }
}

/* one line per column up to the first column without bases.  Old format: the all-bases summary, the five class counts and the
 * first column's count; -N: the first column's count, then the summary of every class */
static void report(FILE *out, int per_class)
{
    if (per_class) {
        fputs("cycle\tmax_count", out);
        for (int cls = 0; cls < CLASSES; ++cls)
            for (int f = 0; f < 11; ++f) fprintf(out, "\t%s_%s", class_name[cls], column_name[f]);
    } else {
        fputs("column", out);
        for (int f = 0; f < 11; ++f) fprintf(out, "\t%s", column_name[f]);
        for (int cls = 1; cls < CLASSES; ++cls) fprintf(out, "\t%s_Count", class_name[cls]);
        fputs("\tMax_count", out);
    }
    fputc('\n', out);
    const int first_column_bases = cell(record(0, 0) + F_BASES);
    for (size_t column = 0; column < MAX_SEQ_LINE_LENGTH && cell(record(column, 0) + F_BASES) != 0; ++column) {
        fprintf(out, "%d", (int)column + 1);
        if (per_class) fprintf(out, "\t%d", first_column_bases);
        for (int cls = 0; cls < (per_class ? CLASSES : 1); ++cls) {
            const struct summary s = summarise(column, cls);
            fputc('\t', out);
            print_summary(out, &s);
        }
        if (!per_class) {
            for (int cls = 1; cls < CLASSES; ++cls) fprintf(out, "\t%d", cell(record(column, cls) + F_BASES));
            fprintf(out, "\t%d", first_column_bases);
        }
        fputc('\n', out);
    }
}

static int want_per_class;
static int option(int index, int letter, char *value)
{
    (void)index; (void)value;
    if (letter != 'N') errx(1, "fastx_quality_stats.c:%d: Unknown argument (%c)", __LINE__, letter);
    want_per_class = 1;
    return 1;
}

int main(int argc, char *argv[])
{
    static FASTX fx;
    fastx_parse_cmdline(argc, argv, "N", option);
    fastx_init_reader(&fx, get_input_filename(), FASTA_OR_FASTQ, ALLOW_N, REQUIRE_UPPERCASE, get_fastq_ascii_quality_offset());
    FILE *out = stdout;
    if (strcmp(get_output_filename(), "-") != 0 && !(out = fopen(get_output_filename(), "w+")))
        err(1, "Failed to create output file (%s)", get_output_filename());
    if (fx.read_fastq) {
        uint64_t *hist = NULL;
        uint32_t cols = 0;
        fxh_totals totals;
        fxh_run_quality_stats(&fx, &hist, &cols, &totals);
        take_histogram(hist, cols);
        free(hist);
    } else take_fasta(&fx);
    report(out, want_per_class);
    fflush(out);
    return 0;
}
