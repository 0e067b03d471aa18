/* fastx_quality_stats -- same command line and output (old and -N format) as the reference tool
 * (src/fastx_quality_stats/fastx_quality_stats.c).  For FASTQ input the per-column histograms are built on the GPU
 * (fxg_run_quality_stats) and everything the reference derives from its counting-sort arrays -- count, min, max, sum,
 * quartiles, whiskers (:218-414) -- is derived here from that histogram with the reference's own arithmetic.  FASTA input
 * (counts only, weighted by collapsed-read multiplicity) stays on the record API. */
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

enum { ALL = 0, NUC_INDEX_SIZE = 6 };
static const char *nucleotide_index_name[NUC_INDEX_SIZE] = {"ALL", "A", "C", "G", "T", "N"};

/* fastx_quality_stats.c:115-134.  The reference declares these after including fastx.h, which leaves #pragma pack(1) on
 * (fastx.h:61), and keeps them in one static array; get_nth_value (:237-243) walks past the end of bases_values_count when a
 * class has bases but no qualities (FASTA input) and lands on the next record's min = 100.  Same layout here, so the same
 * walk reads the same numbers (and, like there, a quality of 93 indexes one past the array: SURVEY N2). */
#pragma pack(push, 1)
struct nucleotide_data {
    int min, max, count;
    unsigned long long sum;
    int bases_values_count[QUALITY_VALUES_RANGE];
};
#pragma pack(pop)
#define MAX_SEQUENCE_LENGTH MAX_SEQ_LINE_LENGTH
static struct nucleotide_data cycles[MAX_SEQUENCE_LENGTH][NUC_INDEX_SIZE];
static size_t ncycles = MAX_SEQUENCE_LENGTH;
static FILE *outfile;
static int new_output_format = 0;
static FASTX fastx;

static void init_values(void)            /* :138-163 */
{
    for (size_t i = 0; i < MAX_SEQUENCE_LENGTH; ++i)
        for (int j = 0; j < NUC_INDEX_SIZE; ++j) { cycles[i][j].min = 100; cycles[i][j].max = -100; }
}
static void grow_cycles(size_t n)
{
    if (n > MAX_SEQUENCE_LENGTH) errx(1, "Internal error: sequence too long. Hard-coded max. length is %d", MAX_SEQ_LINE_LENGTH);
}

static int nuc_to_index(int c)            /* :142-155 */
{
    switch (c) {
    case 'A': case 'a': return 1; case 'C': case 'c': return 2; case 'G': case 'g': return 3;
    case 'T': case 't': return 4; case 'N': case 'n': return 5; default: return 0;
    }
}

/* device histogram [col][A,C,G,T,N][quality + 33] -> the reference's per-cycle records (what read_file :166-216 leaves) */
static void cycles_from_histogram(const uint64_t *hist, uint32_t cols)
{
    grow_cycles(cols);
    for (uint32_t c = 0; c < cols; ++c)
        for (int k = 1; k < NUC_INDEX_SIZE; ++k) {
            const uint64_t *h = hist + ((size_t)c * FXG_QS_CLASSES + (size_t)(k - 1)) * FXG_QS_BINS;
            for (int b = 0; b < FXG_QS_BINS; ++b) {
                if (!h[b]) continue;
                const int v = b - 33;
                struct nucleotide_data *d[2] = {&cycles[c][ALL], &cycles[c][k]};
                for (int t = 0; t < 2; ++t) {
                    if (v < d[t]->min) d[t]->min = v;
                    if (v > d[t]->max) d[t]->max = v;
                    d[t]->count += (int)h[b];
                    d[t]->sum += (unsigned long long)((long long)v * (long long)h[b]);
                    d[t]->bases_values_count[v - MIN_QUALITY_VALUE] += (int)h[b];
                }
            }
        }
}

static void read_fasta_on_host(void)     /* read_file :166-216 without qualities */
{
    while (fastx_read_next_record(&fastx)) {
        const size_t L = strlen(fastx.nucleotides);
        grow_cycles(L);
        const int reads_count = get_reads_count(&fastx);
        for (size_t i = 0; i < L; ++i) {
            cycles[i][ALL].count += reads_count;
            cycles[i][nuc_to_index(fastx.nucleotides[i])].count += reads_count;
        }
    }
}

static int get_nth_value(size_t cycle, int nucleotide, int n)   /* :218-247 */
{
    const struct nucleotide_data *d = &cycles[cycle][nucleotide];
    if (n == 0) return d->min;
    if (n < 0 || n >= d->count) {
        fprintf(stderr, "Internal error at get_nth_value (cycle=%d, nucleotide=%d, n=%d), count_values[%d]=%d\n", (int)cycle, nucleotide, n, (int)cycle, d->count);
        exit(1);
    }
    /* the walk may leave this class' array (FASTA input: bases but no qualities) and run through its neighbours -- as in the reference,
     * whose numbers it then reproduces -- but not past the end of `cycles`: there the reference reads whatever globals follow, this stops */
    const int last = (int)(((const char *)cycles + sizeof cycles - (const char *)d->bases_values_count) / (ptrdiff_t)sizeof(int)) - 1;
    int pos = 0;
    while (n > 0) {
        if (d->bases_values_count[pos] > n) break;
        n -= d->bases_values_count[pos];
        if (pos >= last) break;
        pos++;
        while (d->bases_values_count[pos] == 0 && pos < last) pos++;
    }
    return pos + MIN_QUALITY_VALUE;
}

static void box(size_t cycle, int nuc, int *Q1, int *Q3, int *IQR, int *lw, int *rw)   /* :276-291 */
{
    const struct nucleotide_data *d = &cycles[cycle][nuc];
    *Q1 = get_nth_value(cycle, nuc, d->count / 4);
    *Q3 = get_nth_value(cycle, nuc, d->count * 3 / 4);
    *IQR = *Q3 - *Q1;
    *lw = (*Q1 - *IQR * 3 / 2) < d->min ? d->min : (*Q1 - *IQR * 3 / 2);
    *rw = (*Q3 + *IQR * 3 / 2) > d->max ? d->max : (*Q3 + *IQR * 3 / 2);
}

static void print_nucleotide_statistics(size_t cycle, int nuc)   /* :271-294 */
{
    const struct nucleotide_data *d = &cycles[cycle][nuc];
    int Q1, Q3, IQR, lw, rw;
    box(cycle, nuc, &Q1, &Q3, &IQR, &lw, &rw);
    fprintf(outfile, "\t%d\t%d\t%d\t%lld\t", d->count, d->min, d->max, (long long)d->sum);
    fprintf(outfile, "%3.2f\t%d\t%d\t%d\t", ((double)d->sum) / ((double)d->count), Q1, get_nth_value(cycle, nuc, d->count / 2), Q3);
    fprintf(outfile, "%d\t%d\t%d", IQR, lw, rw);
}

static void print_statistics(void)       /* :296-334 */
{
    static const char *headers[] = {"count", "min", "max", "sum", "mean", "Q1", "med", "Q3", "IQR", "lW", "rW"};
    fprintf(outfile, "cycle\tmax_count");
    for (int nuc = 0; nuc < NUC_INDEX_SIZE; ++nuc)
        for (int h = 0; h < 11; ++h) fprintf(outfile, "\t%s_%s", nucleotide_index_name[nuc], headers[h]);
    fprintf(outfile, "\n");
    const int max_count = ncycles ? cycles[0][ALL].count : 0;
    for (size_t cycle = 0; cycle < ncycles; ++cycle) {
        if (cycles[cycle][ALL].count == 0) break;
        fprintf(outfile, "%d\t%d", (int)cycle + 1, max_count);
        for (int nuc = 0; nuc < NUC_INDEX_SIZE; ++nuc) print_nucleotide_statistics(cycle, nuc);
        fprintf(outfile, "\n");
    }
}

static void print_old_statistics(void)   /* :340-414 */
{
    fprintf(outfile, "column\t");
    fprintf(outfile, "count\tmin\tmax\tsum\t");
    fprintf(outfile, "mean\tQ1\tmed\tQ3\t");
    fprintf(outfile, "IQR\tlW\trW\t");
    fprintf(outfile, "A_Count\tC_Count\tG_Count\tT_Count\tN_Count\t");
    fprintf(outfile, "Max_count\n");
    for (size_t i = 0; i < ncycles; ++i) {
        const struct nucleotide_data *d = &cycles[i][ALL];
        if (d->count == 0) break;
        int Q1, Q3, IQR, lw, rw;
        box(i, ALL, &Q1, &Q3, &IQR, &lw, &rw);
        fprintf(outfile, "%d\t", (int)i + 1);
        fprintf(outfile, "%d\t%d\t%d\t%lld\t", d->count, d->min, d->max, (long long)d->sum);
        fprintf(outfile, "%3.2f\t%d\t%d\t%d\t", ((double)d->sum) / ((double)d->count), Q1, get_nth_value(i, ALL, d->count / 2), Q3);
        fprintf(outfile, "%d\t%d\t%d\t", IQR, lw, rw);
        fprintf(outfile, "%d\t%d\t%d\t%d\t%d\t", cycles[i][1].count, cycles[i][2].count, cycles[i][3].count, cycles[i][4].count, cycles[i][5].count);
        fprintf(outfile, "%d\n", cycles[0][ALL].count);
    }
}

static int parse_program_args(int optind_, int optc, char *optarg_)   /* :417-428 */
{
    (void)optind_; (void)optarg_;
    switch (optc) {
    case 'N': new_output_format = 1; break;
    default: errx(1, "fastx_quality_stats.c:%d: Unknown argument (%c)", __LINE__, optc);
    }
    return 1;
}

int main(int argc, char *argv[])
{
    fastx_parse_cmdline(argc, argv, "N", parse_program_args);                       /* :432-447 */
    init_values();
    fastx_init_reader(&fastx, get_input_filename(), FASTA_OR_FASTQ, ALLOW_N, REQUIRE_UPPERCASE, get_fastq_ascii_quality_offset());
    if (strcmp(get_output_filename(), "-") == 0) outfile = stdout;
    else {
        outfile = fopen(get_output_filename(), "w+");
        if (outfile == NULL) err(1, "Failed to create output file (%s)", get_output_filename());
    }
    if (fastx.read_fastq) {
        uint64_t *hist = NULL;
        uint32_t cols = 0;
        fxh_totals tot;
        fxh_run_quality_stats(&fastx, &hist, &cols, &tot);
        cycles_from_histogram(hist, cols);
        free(hist);
    } else read_fasta_on_host();
    if (new_output_format) print_statistics(); else print_old_statistics();
    fflush(outfile);
    return 0;
}
