//go:build withfhe

// What lives in package fhe: the ciphertext-axis transform (fhe.NTT, fhe/ntt.go:12-281), Commit's root
// and a whole EncryptedProof.MarshalBinary at a tiny shape.  Needs the lazer libraries (package fhe
// imports vdec, fhe/ligero.go:14):   go run -tags withfhe ./cmd/lumen_dump -out DIR
package main

import (
	"fmt"

	"github.com/nulltea/lumenos/core"
	"github.com/nulltea/lumenos/fhe"
	"github.com/tuneinsight/lattigo/v6/core/rlwe"
	"github.com/tuneinsight/lattigo/v6/schemes/bgv"
)

func dumpFHE(out string, params bgv.Parameters, sk *rlwe.SecretKey, pk *rlwe.PublicKey, encoder *bgv.Encoder,
	encryptor *rlwe.Encryptor) {
	const rows, cols, rhoInv = 64, 16, 2
	kgen := rlwe.NewKeyGenerator(params)
	rlk := kgen.GenRelinearizationKeyNew(sk)
	galEls := params.GaloisElementsForInnerSum(1, rows)
	gks := kgen.GenGaloisKeysNew(galEls, sk)
	evk := rlwe.NewMemEvaluationKeySet(rlk, gks...)
	ptField, err := core.NewPrimeField(params.PlaintextModulus(), cols*rhoInv)
	must(err)
	server := fhe.NewBackendBFV(&ptField, params, pk, evk)

	_, batched, err := core.RandomMatrixRowMajor(rows, cols, Modulus, func(u []uint64) *rlwe.Plaintext {
		pt := bgv.NewPlaintext(params, params.MaxLevel())
		must(encoder.Encode(u, pt))
		return pt
	})
	must(err)
	cts := make([]*rlwe.Ciphertext, cols)
	for i, pt := range batched {
		cts[i], err = server.EncryptNew(pt)
		must(err)
	}

	// ---- ct_ntt.lmfx: fhe.NTT in place on 2*cols ciphertexts (deterministic given its inputs)
	{
		f := newFile()
		vals := make([]*rlwe.Ciphertext, 0, cols*rhoInv)
		for i := 0; i < cols*rhoInv; i++ {
			vals = append(vals, cts[i%cols].CopyNew())
		}
		for i, c := range vals {
			f.ct(fmt.Sprintf("in%d", i), c)
		}
		res, err := fhe.NTT(vals, cols*rhoInv, server)
		must(err)
		for i, c := range res {
			f.ct(fmt.Sprintf("out%d", i), c)
		}
		f.scalar("mul_counter", uint64(server.MulCounter()))
		f.save(out, "ct_ntt.lmfx")
	}

	// ---- proof.lmfx: Commit + Prove + MarshalBinary (fhe/ligero.go:95-291, 646-705)
	{
		f := newFile()
		ligero, err := fhe.NewLigeroCommitter(128, rows, cols, rhoInv)
		must(err)
		span := core.StartSpan("dump", nil)
		prover, root, err := ligero.Commit(cts, server, span)
		must(err)
		f.raw("root", root)
		for i, c := range prover.EncodedMatrix {
			f.ct(fmt.Sprintf("encoded%d", i), c) // top level; the leaves hash their level-1 rescalings
		}
		transcript := core.NewTranscript("test")
		z := core.NewElement(1)
		proof, err := prover.Prove(z, server, transcript, span)
		must(err)
		for i, c := range proof.MatR {
			f.ct(fmt.Sprintf("matR%d", i), c)
		}
		for i, c := range proof.QueriedCols {
			f.ct(fmt.Sprintf("queried%d", i), c)
		}
		b, err := proof.MarshalBinary()
		must(err)
		f.raw("marshaled", b)
		f.polysQP("sk", sk.Value)
		for i, gk := range gks {
			f.scalar(fmt.Sprintf("key%d.galois_element", i), gk.GaloisElement)
			f.evk(fmt.Sprintf("key%d", i), &gk.EvaluationKey)
		}
		for i, c := range cts {
			f.ct(fmt.Sprintf("matrix%d", i), c)
		}
		f.save(out, "proof.lmfx")
	}
}
