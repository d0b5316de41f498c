// lumen_dump -- writes the Lattigo-side fixtures that pin this repository's CPU oracle (and through it
// the HIP kernels) against the reference's own arithmetic.
//
// NOT BUILT HERE: the build image has no Go toolchain and no module cache (DESIGN.md section 4), so this
// program has never been compiled; it is written against the API the reference itself uses
// (fhe/*.go, core/field.go, cmd/*/main.go name every type and method touched below) at the pinned
//     github.com/tuneinsight/lattigo/v6 v6.1.2-0.20250520151126-84f6bc33cb5b        (go.mod:5)
// A maintainer with a Go host runs it ONCE:
//
//     cp -r tools/go_dump  <lumenos checkout>/cmd/lumen_dump
//     cd <lumenos checkout> && go run ./cmd/lumen_dump -out /tmp/lattigo_fixtures
//     cp /tmp/lattigo_fixtures/*.lmfx  <this repo>/tests/golden/lattigo/
//     python -m pytest tests/test_lattigo_fixtures.py            # CPU: oracle vs Lattigo
//     python -m pytest tests/test_lattigo_fixtures.py -m gpu     # GPU: HIP vs Lattigo
//
// It imports only Lattigo and the reference's pure-Go `core` package, so it needs neither lazer nor cgo.
// The optional file fhe_dump.go (build tag `withfhe`) adds what lives in package fhe (fhe.NTT on
// ciphertexts, a whole EncryptedProof.MarshalBinary) and therefore needs the lazer libraries on the
// linker path (README.md:21-24 of the reference).
//
// Every record is what tests/lmfx.py reads: name, kind, dims, little-endian payload.  Polynomials are
// dumped EXACTLY AS LATTIGO STORES THEM (the flags say whether that is NTT / Montgomery form); the
// Python side converts.  Layout of a ciphertext record: [poly][limb][N].
package main

import (
	"bytes"
	"encoding/binary"
	"encoding/json"
	"flag"
	"fmt"
	"os"
	"path/filepath"

	"github.com/nulltea/lumenos/core"
	"github.com/tuneinsight/lattigo/v6/core/rlwe"
	"github.com/tuneinsight/lattigo/v6/ring"
	"github.com/tuneinsight/lattigo/v6/ring/ringqp"
	"github.com/tuneinsight/lattigo/v6/schemes/bgv"
)

const Modulus = 144115188075593729 // cmd/server/main.go:22

// ---------------------------------------------------------------- LMFX1 writer
type lmfx struct{ buf bytes.Buffer }

func newFile() *lmfx {
	f := &lmfx{}
	f.buf.WriteString("LMFX1\x00\x00\x00")
	return f
}

func (f *lmfx) header(name string, kind uint8, dims ...uint64) {
	binary.Write(&f.buf, binary.LittleEndian, uint32(len(name)))
	f.buf.WriteString(name)
	binary.Write(&f.buf, binary.LittleEndian, kind)
	binary.Write(&f.buf, binary.LittleEndian, uint32(len(dims)))
	for _, d := range dims {
		binary.Write(&f.buf, binary.LittleEndian, d)
	}
}

func (f *lmfx) u64s(name string, v []uint64) {
	f.header(name, 8, uint64(len(v)))
	binary.Write(&f.buf, binary.LittleEndian, v)
}

func (f *lmfx) scalar(name string, v uint64) { f.u64s(name, []uint64{v}) }

func (f *lmfx) raw(name string, b []byte) {
	f.header(name, 1, uint64(len(b)))
	f.buf.Write(b)
}

// polys: [len(polys)][limbs][N]
func (f *lmfx) polys(name string, polys ...ring.Poly) {
	limbs := len(polys[0].Coeffs)
	n := len(polys[0].Coeffs[0])
	f.header(name, 8, uint64(len(polys)), uint64(limbs), uint64(n))
	for _, p := range polys {
		for _, limb := range p.Coeffs {
			binary.Write(&f.buf, binary.LittleEndian, limb)
		}
	}
}

// a ciphertext with its metadata flags
func (f *lmfx) ct(name string, ct *rlwe.Ciphertext) {
	f.polys(name, ct.Value...)
	flags := []uint64{0, 0, 0, uint64(ct.Level())}
	if ct.IsNTT {
		flags[0] = 1
	}
	if ct.IsMontgomery {
		flags[1] = 1
	}
	if ct.IsBatched {
		flags[2] = 1
	}
	f.u64s(name+".flags", flags) // IsNTT, IsMontgomery, IsBatched, level
	f.scalar(name+".scale", ct.Scale.Uint64())
}

// polys over QP: Q limbs then P limbs of each -> [len][L+K][N]
func (f *lmfx) polysQP(name string, polys ...ringqp.Poly) {
	limbs := len(polys[0].Q.Coeffs) + len(polys[0].P.Coeffs)
	n := len(polys[0].Q.Coeffs[0])
	f.header(name, 8, uint64(len(polys)), uint64(limbs), uint64(n))
	for _, p := range polys {
		for _, limb := range p.Q.Coeffs {
			binary.Write(&f.buf, binary.LittleEndian, limb)
		}
		for _, limb := range p.P.Coeffs {
			binary.Write(&f.buf, binary.LittleEndian, limb)
		}
	}
}

func (f *lmfx) save(dir, name string) {
	path := filepath.Join(dir, name)
	if err := os.WriteFile(path, f.buf.Bytes(), 0o644); err != nil {
		panic(err)
	}
	fmt.Printf("wrote %s (%d bytes)\n", path, f.buf.Len())
}

func must(err error) {
	if err != nil {
		panic(err)
	}
}

// standard-form psi of a SubRing: RootsForward[N/2] = MForm(psi^bitrev(N/2)) = MForm(psi); MRed(x, 1) = x * 2^-64
func psiOf(s *ring.SubRing) uint64 {
	return ring.MRed(s.RootsForward[s.N/2], 1, s.Modulus, s.MRedConstant)
}

// evaluation key: [digitRNS][digitPow2][b|a] of polys over QP, flattened digit-major -> [D*D2*2][L+K][N]
func (f *lmfx) evk(name string, evk *rlwe.EvaluationKey) {
	var all []ringqp.Poly
	for i := range evk.Value {
		for j := range evk.Value[i] {
			all = append(all, evk.Value[i][j][0], evk.Value[i][j][1])
		}
	}
	f.polysQP(name, all...)
	f.u64s(name+".shape", []uint64{uint64(len(evk.Value)), uint64(len(evk.Value[0])), uint64(evk.BaseTwoDecomposition)})
}

func main() {
	out := flag.String("out", "lattigo_fixtures", "output directory")
	logN := flag.Int("logN", 10, "ring degree of the fixtures")
	cols := flag.Int("cols", 16, "matrix columns (chooses the Q chain as fhe.GenerateBGVParamsForNTT does)")
	keySizes := flag.Bool("keysizes", false, "also marshal the whole KeysRequest of cmd/client (run with -logN 12 -cols 1024 to reproduce '69 MB')")
	flag.Parse()
	must(os.MkdirAll(*out, 0o755))

	// fhe.GenerateBGVParamsForNTT (fhe/bfv.go:121-188) restated for T > 45 bits so that package fhe
	// (and with it lazer) is not needed: LogQ = [58, 56 x (log2(cols) - 1)], LogP = [55, 55]
	k := 0
	for (1 << k) < *cols {
		k++
	}
	logQ := []int{58}
	for i := 1; i < k; i++ {
		logQ = append(logQ, 56)
	}
	params, err := bgv.NewParametersFromLiteral(bgv.ParametersLiteral{
		LogN: *logN, LogQ: logQ, LogP: []int{55, 55}, PlaintextModulus: Modulus,
	})
	must(err)
	N := params.N()
	L := len(params.Q())
	ringQ := params.RingQ()

	kgen := rlwe.NewKeyGenerator(params)
	sk, pk := kgen.GenKeyPairNew()
	encoder := bgv.NewEncoder(params)
	encryptor := rlwe.NewEncryptor(params, pk)
	decryptor := rlwe.NewDecryptor(params, sk)

	// ---- params.lmfx: moduli, roots, field table
	{
		f := newFile()
		f.scalar("logN", uint64(*logN))
		f.u64s("Q", params.Q())
		f.u64s("P", params.P())
		f.scalar("T", params.PlaintextModulus())
		var psi []uint64
		for _, s := range ringQ.SubRings {
			psi = append(psi, psiOf(s))
		}
		for _, s := range params.RingP().SubRings {
			psi = append(psi, psiOf(s))
		}
		f.u64s("psi", psi)
		f.scalar("psi_T", psiOf(params.RingT().SubRings[0]))
		f.u64s("roots_forward_q0_montgomery", ringQ.SubRings[0].RootsForward)
		// the list the CLIENT generates keys for (fhe/ligero_test.go:53, cmd/client/main.go:81): expected
		// rotations {1..n/2, n} plus the row swap iff n > N/2 -- log2(n)+1 resp. log2(N)+2 elements, in Go's map order
		f.u64s("galois_elements_inner_sum_half", params.GaloisElementsForInnerSum(1, N/2))
		f.u64s("galois_elements_inner_sum_full", params.GaloisElementsForInnerSum(1, N))
		f.scalar("galois_count_half", uint64(len(params.GaloisElementsForInnerSum(1, N/2))))
		f.scalar("galois_count_full", uint64(len(params.GaloisElementsForInnerSum(1, N))))
		f.scalar("galois_element_row_swap", params.GaloisElementForRowRotation())
		// serialised sizes behind "Marshaled keys length" (cmd/client/main.go:87-132): pk, rlk, one Galois key
		pkBytes, err := pk.MarshalBinary()
		must(err)
		rlk := kgen.GenRelinearizationKeyNew(sk)
		rlkBytes, err := rlk.MarshalBinary()
		must(err)
		gk0 := kgen.GenGaloisKeyNew(params.GaloisElement(1), sk)
		gkBytes, err := gk0.MarshalBinary()
		must(err)
		f.scalar("pk_marshal_len", uint64(len(pkBytes)))
		f.scalar("rlk_marshal_len", uint64(len(rlkBytes)))
		f.scalar("galois_key_marshal_len", uint64(len(gkBytes)))
		f.u64s("rlk.shape", []uint64{uint64(len(rlk.Value)), uint64(len(rlk.Value[0])), uint64(rlk.BaseTwoDecomposition)})
		if *keySizes { // the whole KeysRequest as the client posts it, for n = N/2 and n = N (heavy at real sizes)
			for _, n := range []int{N / 2, N} {
				gks := kgen.GenGaloisKeysNew(params.GaloisElementsForInnerSum(1, n), sk)
				rot := make([][]byte, len(gks))
				for i, k := range gks {
					rot[i], err = k.MarshalBinary()
					must(err)
				}
				body, err := json.Marshal(struct {
					PublicKey          []byte                 `json:"public_key"`
					RelinearizationKey []byte                 `json:"relinearization_key"`
					RotationKeys       [][]byte               `json:"rotation_keys"`
					RingSwitchEvk      []byte                 `json:"ring_switch_evk"`
					ParamsLit          *bgv.ParametersLiteral `json:"params_lit"`
				}{PublicKey: pkBytes, RelinearizationKey: rlkBytes, RotationKeys: rot})
				must(err)
				f.scalar(fmt.Sprintf("keys_request_len_n%d", n), uint64(len(body)))
				fmt.Printf("Marshaled keys length (rows = %d): %d bytes\n", n, len(body))
			}
		}
		field, err := core.NewPrimeField(params.PlaintextModulus(), 2*(*cols))
		must(err)
		roots := make([]uint64, field.N())
		for i := range roots {
			roots[i] = field.RootForwardUint64(i)
		}
		f.u64s("field_roots_forward", roots) // backend.Field().RootForwardUint64(i), core/field.go:45-47
		f.save(*out, "params.lmfx")
	}

	values := make([]uint64, N)
	for i := range values {
		values[i] = (uint64(i)*0x9e3779b97f4a7c15 + 12345) % Modulus
	}
	pt := bgv.NewPlaintext(params, params.MaxLevel())
	must(encoder.Encode(values, pt))
	ct, err := encryptor.EncryptNew(pt)
	must(err)

	// ---- encrypt.lmfx: keys, Encoder.Encode, EncryptNew (randomised: pins decryption + noise size)
	{
		f := newFile()
		f.u64s("values", values)
		f.polys("plaintext", pt.Value) // NTT domain, m * T^-1 form
		f.polysQP("sk", sk.Value)      // Lattigo keeps keys in NTT + Montgomery form
		f.polysQP("pk", pk.Value[0], pk.Value[1])
		f.scalar("keys_montgomery", 1)
		f.ct("ciphertext", ct)
		dec := make([]uint64, N)
		must(encoder.Decode(decryptor.DecryptNew(ct), dec))
		f.u64s("decrypted", dec)
		f.save(*out, "encrypt.lmfx")
	}

	eval := bgv.NewEvaluator(params, nil)

	// ---- mul_scalar.lmfx: Evaluator.Mul(ct, uint64) with a table word above and one below T/2 (SURVEY A.2)
	{
		f := newFile()
		f.ct("in", ct)
		for i, w := range []uint64{95661681840738641, 33554304, Modulus - 1, 3} {
			o := ct.CopyNew()
			must(eval.Mul(ct, w, o))
			f.scalar(fmt.Sprintf("w%d", i), w)
			f.ct(fmt.Sprintf("out%d", i), o)
		}
		a, b := ct.CopyNew(), ct.CopyNew()
		must(eval.Add(ct, ct, a))
		must(eval.Sub(a, ct, b))
		f.ct("add", a)
		f.ct("sub", b)
		f.save(*out, "mul_scalar.lmfx")
	}

	// ---- mul_plain.lmfx: MulNew(ct, pt) (fhe/ligero.go:319)
	r := make([]uint64, N)
	for i := range r {
		r[i] = uint64(i)*0xbf58476d1ce4e5b9 + 7 // raw u64, not reduced mod T (ligero.go:202-205)
	}
	rPt := bgv.NewPlaintext(params, params.MaxLevel())
	must(encoder.Encode(r, rPt))
	{
		f := newFile()
		f.ct("in", ct)
		f.u64s("r", r)
		f.polys("plaintext", rPt.Value)
		o, err := eval.MulNew(ct, rPt)
		must(err)
		f.ct("out", o)
		f.save(*out, "mul_plain.lmfx")
	}

	// ---- rescale.lmfx: one Rescale, then the loop `for ct.Level() > 1` (fhe/ligero.go:149-155)
	var level1 *rlwe.Ciphertext
	{
		f := newFile()
		f.ct("in", ct)
		o := ct.CopyNew()
		must(eval.Rescale(o, o))
		f.ct("once", o)
		for o.Level() > 1 {
			must(eval.Rescale(o, o))
		}
		f.ct("level1", o)
		level1 = o
		f.save(*out, "rescale.lmfx")
	}

	// ---- writeto.lmfx: the leaf bytes of Commit (ct.WriteTo, fhe/ligero.go:156-157)
	{
		f := newFile()
		f.ct("ct", level1)
		var b bytes.Buffer
		_, err := level1.WriteTo(&b)
		must(err)
		f.raw("bytes", b.Bytes())
		md, err := level1.MetaData.MarshalBinary()
		must(err)
		f.raw("metadata_bytes", md)
		f.save(*out, "writeto.lmfx")
	}

	// ---- innersum.lmfx: MulNew + InnerSum(ct, 1, n) + Rescale loop for n = N/2 and n = N
	// (matrixInnerSumEval, fhe/ligero.go:318-333; n = N is what configs B, C, D run -- SURVEY Appendix D-1)
	for _, n := range []int{N / 2, N} {
		galEls := params.GaloisElementsForInnerSum(1, n)
		gks := kgen.GenGaloisKeysNew(galEls, sk)
		ev := bgv.NewEvaluator(params, rlwe.NewMemEvaluationKeySet(nil, gks...))
		f := newFile()
		f.scalar("n", uint64(n))
		f.u64s("galois_elements", galEls)
		for i, gk := range gks {
			f.scalar(fmt.Sprintf("key%d.galois_element", i), gk.GaloisElement)
			f.evk(fmt.Sprintf("key%d", i), &gk.EvaluationKey)
		}
		f.scalar("keys_montgomery", 1)
		f.polysQP("sk", sk.Value)
		f.ct("in", ct)
		f.u64s("r", r)
		f.polys("plaintext", rPt.Value)
		col, err := ev.MulNew(ct, rPt)
		must(err)
		if n == N {
			// where does the row swap sit?  Both orders decrypt to the same slots but differ bit for bit:
			//   A: InnerSum over the N/2 columns first, then + RotateRows        (what this repository restates)
			//   B: ct + RotateRows(ct) first, then InnerSum over the N/2 columns
			// plus the intermediate after the last column rotation of A
			a := col.CopyNew()
			must(ev.InnerSum(a, 1, N/2, a))
			f.ct("cols_only", a)
			ar, err := ev.RotateRowsNew(a)
			must(err)
			f.ct("rows_of_cols", ar)
			must(ev.Add(a, ar, a))
			f.ct("cand_cols_then_rows", a)
			b := col.CopyNew()
			br, err := ev.RotateRowsNew(b)
			must(err)
			must(ev.Add(b, br, b))
			must(ev.InnerSum(b, 1, N/2, b))
			f.ct("cand_rows_then_cols", b)
		}
		must(ev.InnerSum(col, 1, n, col))
		f.ct("inner_sum", col)
		for col.Level() > 1 {
			must(ev.Rescale(col, col))
		}
		f.ct("out", col)
		dec := make([]uint64, 1)
		must(encoder.Decode(decryptor.DecryptNew(col), dec))
		f.u64s("slot0", dec)
		f.save(*out, fmt.Sprintf("innersum_%d.lmfx", n))
	}

	// ---- ringswitch.lmfx: fhe/ring_switch.go:16-57 (key) and :106-113 (ApplyEvaluationKey into the small ring)
	{
		small := 8
		if *logN > 10 {
			small = 10
		}
		paramsNew, err := bgv.NewParametersFromLiteral(bgv.ParametersLiteral{
			LogN: small, Q: []uint64{params.Q()[0]}, P: []uint64{}, PlaintextModulus: params.PlaintextModulus(),
		})
		must(err)
		skNew := rlwe.NewKeyGenerator(paramsNew).GenSecretKeyNew()
		lvlQ, lvlP, base := params.MaxLevel(), params.MaxLevelP(), 13
		rsEvk := kgen.GenEvaluationKeyNew(sk, skNew, rlwe.EvaluationKeyParameters{
			LevelQ: &lvlQ, LevelP: &lvlP, BaseTwoDecomposition: &base,
		})
		f := newFile()
		f.scalar("logN_small", uint64(small))
		f.evk("key", rsEvk)
		f.scalar("keys_montgomery", 1)
		f.ct("in", level1)
		ct2 := rlwe.NewCiphertext(paramsNew, 1, paramsNew.MaxLevel())
		must(eval.ApplyEvaluationKey(level1, rsEvk, ct2))
		f.ct("out", ct2)
		f.polysQP("sk_small", skNew.Value)
		f.scalar("level_p", uint64(lvlP))
		rsBytes, err := rsEvk.MarshalBinary()
		must(err)
		f.scalar("key_marshal_len", uint64(len(rsBytes))) // the "+ 5 / 7 / 15 / 29 MB" of the experimental key-size logs
		f.save(*out, "ringswitch.lmfx")
	}
	// ---- ringswitch_nop.lmfx: TestRingSwitch's own parameters (fhe/ring_switch_test.go:13-77): LogQ = [58], NO
	// special prime, T = 0x3ee0001, same ring degree; the key then has LevelP = -1 and ApplyEvaluationKey takes
	// the bit-decomposed gadget product (no ModDown)
	{
		p2, err := bgv.NewParametersFromLiteral(bgv.ParametersLiteral{LogN: *logN, LogQ: []int{58}, PlaintextModulus: 0x3ee0001})
		must(err)
		kg2 := rlwe.NewKeyGenerator(p2)
		sk2, pk2 := kg2.GenKeyPairNew()
		enc2 := bgv.NewEncoder(p2)
		m := []uint64{1, 1}
		pt2 := bgv.NewPlaintext(p2, p2.MaxLevel())
		pt2.IsBatched = true
		must(enc2.Encode(m, pt2))
		ct0, err := rlwe.NewEncryptor(p2, pk2).EncryptNew(pt2)
		must(err)
		pNew, err := bgv.NewParametersFromLiteral(bgv.ParametersLiteral{
			LogN: *logN, Q: []uint64{p2.Q()[0]}, P: []uint64{}, PlaintextModulus: p2.PlaintextModulus(),
		})
		must(err)
		skNew := rlwe.NewKeyGenerator(pNew).GenSecretKeyNew()
		lvlQ, lvlP, base := p2.MaxLevel(), p2.MaxLevelP(), 13
		rsEvk := kg2.GenEvaluationKeyNew(sk2, skNew, rlwe.EvaluationKeyParameters{LevelQ: &lvlQ, LevelP: &lvlP, BaseTwoDecomposition: &base})
		f := newFile()
		f.scalar("logN", uint64(*logN))
		f.u64s("Q", p2.Q())
		f.scalar("T", p2.PlaintextModulus())
		f.scalar("psi_q0", psiOf(p2.RingQ().SubRings[0]))
		f.scalar("level_p", uint64(int64(lvlP))) // -1 as two's complement
		f.evk("key", rsEvk)
		f.scalar("keys_montgomery", 1)
		f.ct("in", ct0)
		ct2 := rlwe.NewCiphertext(pNew, 1, pNew.MaxLevel())
		must(bgv.NewEvaluator(p2, nil).ApplyEvaluationKey(ct0, rsEvk, ct2))
		f.ct("out", ct2)
		f.polys("sk_new", skNew.Value.Q)
		dec := make([]uint64, len(m))
		must(bgv.NewEncoder(pNew).Decode(rlwe.NewDecryptor(pNew, skNew).DecryptNew(ct2), dec))
		f.u64s("decoded", dec)
		f.save(*out, "ringswitch_nop.lmfx")
	}
	_ = L
	dumpFHE(*out, params, sk, pk, encoder, encryptor) // no-op unless built with -tags withfhe
}
